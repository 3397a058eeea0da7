// Winograd F(2x2, 3x3) convolution for gfx950 on v_mfma_f32_16x16x4_f32: the 3x3 / stride 1 / pad 1 layers with wide
// channel counts (the residual trunk: 256 -> 256 at H/4 x W/4, 57 % of the step's multiply-accumulates).
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// turns the 36 multiply-accumulates of a tile and channel pair into 16: sixteen independent GEMMs (one per frequency
// f = 4 fi + fj) of [tiles] x [K] x [N].  Everything stays exact fp32 arithmetic on the matrix pipe; the transforms add
// and subtract only (the halves live in the weight image).  The reference computes these layers with cuDNN, which picks
// the same algorithm family for fp32 3x3 convolutions (scripts/networks.py:695-701 -> nn.Conv2d).
//
// One block = 8x8 tiles (16x16 output pixels) of one image x 64 output channels x all 16 frequencies, 8 waves:
//   * wave w owns frequencies 2w, 2w+1: two 64 x 64 accumulator tiles = 128 registers;
//   * K runs in chunks of 8 channels.  Waves 0-3 are the loaders: per chunk a thread (tile, channel pair) reads its 4x4
//     patch straight from global memory (buffer_load_dwordx2; neighbouring tiles overlap and the 4x re-read is served by
//     the vector L1; zero padding = an out-of-range offset), transforms it in registers (packed adds) and writes 16
//     pairs V[f][tile][k, k+1] to LDS.  Waves 4-7 only multiply, so the matrix pipe of every SIMD is fed by its
//     multiply-only wave while its loader wave transforms.  Measured on the trunk layer (tools/time_conv.py, back-to-back
//     launches): all eight waves loading one channel each, four before and four after their MFMAs: 203 us; without that
//     stagger 213; this split: 195; with U in registers (next item) 188.
//   * The transformed weights U[f][n][k] never touch LDS: the two frequency planes of a wave are private to it, so it reads
//     them from the prepared image (fragment order, L2) straight into registers, four global_load_dwordx4 per chunk,
//     requested one chunk ahead.  (They used to travel global -> LDS with global_load_lds_dwordx4: issuing those cost 11 %
//     of the launch in a timing-only ablation, and interleaving them with the MFMAs made it worse.)
//   * V is double buffered (2 x 32 KiB): one barrier per chunk is the only synchronisation;
//   * epilogue: the 16 frequency planes meet in LDS (two halves of 32 channels), one thread per (tile, channel) folds
//     them into the 2x2 pixels, adds bias, applies the activation and stores 128-byte row segments.
#include "wino.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr int WT = 64;                 // tiles per block
constexpr int WNB = 64;                // output channels per block
constexpr int WK = 8;                  // channels per chunk
constexpr int VBUF = 16 * WT * WK;     // floats per V buffer (32 KiB)
constexpr int UBUF = 16 * WNB * WK;    // floats of U per chunk and N block (32 KiB)
constexpr int MLD = 36;                // row stride of the epilogue's M[f][tile][32] planes: 4*36 = 16 (mod 64) banks
constexpr int WINO_SMEM = 16 * WT * MLD;   // 147 456 B (the epilogue's planes); the V buffers need 65 536
static_assert(WINO_SMEM >= 2 * VBUF, "operand buffers must fit");

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Diagnostic build only (make alt ALTFLAGS=-DWINO_STAMP, tools/wino_stamps.py): per-wave cycle accounting of the main loop with
// s_memtime; nothing of it exists in the product library.
#ifdef WINO_STAMP
__device__ long long g_wino_stamps[8 * 8 * 4];   // [block < 8][wave][segment]
#define WSTAMP(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif

// none / ReLU / LeakyReLU (tanh layers have 3 output channels and never come here)
__device__ inline float wino_act(float v, int act, float slope) {
  return act == MUNIT_ACT_NONE ? v : (v > 0.f ? v : (act == MUNIT_ACT_RELU ? 0.f : v * slope));
}

// MODE 0: reflect padding (forward).  1: zero padding (forward of a zero-padded layer; backward-data of one).
// 2: backward-data of a reflect-padded layer = zero-padded correlation of dy with the rotated filter + the fold of the
//    padded border back onto rows / columns 1 and H-2 / W-2.  The fold needs no second pass: the extra term of output
//    row 1 is dy row 0 under filter row 0, and inside the top tile (output rows 0, 1; patch rows -1 .. 2) filter row 0
//    meets patch row 3 for output row 1 only -- so patch row 3 += patch row 1 there; mirrored at the bottom (patch row
//    0 += patch row 2) and along the columns, the corners get the product of both.  Exact: the transform is linear.
// S2: the 4x4 / stride 2 / pad 1 layers.  Over the four input phases (row and column parity of the padded image) such a
// layer is a sum of four 2x2 / stride 1 VALID convolutions, and F(3x3, 2x2) shares everything with F(2x2, 3x3) but the
// small matrices: the same 4x4 patch, the same B^T, sixteen frequencies -- G = [1 0; 1/2 1/2; 1/2 -1/2; 0 1] and
// A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 -1] (3x3 outputs per tile).  The contraction runs phase-major over K = 4 Cin (the loader
// re-derives its 16 patch offsets when the phase changes); 16 instead of 36 multiply-accumulates per tile again.
// KIND 2: backward-data of such a layer.  Per axis the padded gradient splits by the parity of its index q = 2u + par into
// dxp[u] = sum_a dy[u - a] w[2a + par], a 2-tap correlation over dy: four launch phases (gridDim.y = row, column parity), each an
// F(3x3, 2x2) convolution over the dense dy with the taps reversed and the channel roles swapped, written to every other row /
// column of dx (image row q - 1).  The padded rows q = 0 and q = H + 1 fold (reflect padding) onto image rows 1 and H - 2,
// which sit in the same 3x3 tile of the same phase (the host admits the layer only when H/2 is not a multiple of 3): the
// epilogue adds them in registers before the store; with zero padding they are dropped.
template <int MODE, int KIND = 0>
__global__ __launch_bounds__(512) void conv_wino_kernel(WinoParams p) {
  constexpr bool REFLECT = MODE == 0;
  // KIND 3: backward-data of a sub-pixel up-sampling layer over its interior source pixels: the sum over the four output phases
  // of 3x3 correlations of dy's phase planes (rows 2u + a) with the rotated merged filters -- F(2x2, 3x3) tiles, K = 4 Cout
  // phase-major like KIND 1, no padding anywhere (the 2-pixel frame is the generic kernel's)
  constexpr bool S2 = KIND == 1, DG2 = KIND == 2, UPD = KIND == 3, LIN = KIND == 1 || KIND == 2, PHK = KIND == 1 || KIND == 3;
  __shared__ __attribute__((aligned(16))) float smem[WINO_SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware order (blocks b and b+8 share an XCD): every XCD gets one contiguous run of blocks, so the N-blocks of a
  // tile block and neighbouring tile blocks (shared halo) meet in one L2
  int blk = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blk & 7, idx = blk >> 3;
    blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int n_blk = blk % p.NB;
  int m_blk = blk / p.NB;
  const int m_blk0 = m_blk;
  const int btx = m_blk % p.btw; m_blk /= p.btw;
  const int bty = m_blk % p.bth;
  const int b = m_blk / p.bth;

  // ---- loader (waves 0-3 only): thread = (tile tl, channel pair cp of the chunk); waves 4-7 only multiply ----
#ifdef WINO_SWAP_ROLES      // experiment: the second-dispatched half of the block loads and transforms
  const bool loader = wave >= 4;
#else
  const bool loader = wave < 4;
#endif
#if defined(WINO_PRIO) && WINO_PRIO == 1
  if (!loader) __builtin_amdgcn_s_setprio(1);
#elif defined(WINO_PRIO) && WINO_PRIO == 2
  if (loader) __builtin_amdgcn_s_setprio(1);
#endif
  const int tl = (tid & 255) >> 2, cp = tid & 3;
  // tile of this loader thread, clamped (stores are predicated).  3x3 layers: 8x8 tiles of image b; S2: 64 consecutive tiles
  // of the batch-wide list (b, ty, tx) -- 3x3-pixel tiles rarely divide the extent, and whole 8x8 blocks would waste up to a
  // quarter of the slots
  int gy, gx, gb = b;
  if constexpr (LIN) {
    const int t = min(m_blk0 * 64 + tl, p.B * p.th * p.tw - 1);
    gx = t % p.tw; gy = (t / p.tw) % p.th; gb = t / (p.tw * p.th);
  } else {
    gy = min(bty * 8 + (tl >> 3), p.th - 1); gx = min(btx * 8 + (tl & 7), p.tw - 1);
  }
  unsigned off[16];
  // byte offsets of the thread's 4x4 patch (channel pair cp of the first chunk of a phase); padding that is not a reflection
  // and everything past the image = an offset beyond the buffer, which the load returns as 0
  auto make_off = [&](int phase) {
    int ro[4], co[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ih, iw;
      if constexpr (S2) {   // row q of phase plane r is padded row 2q + r = image row 2q + r - 1
        ih = 2 * (3 * gy + i) + (phase >> 1) - 1;
        iw = 2 * (3 * gx + i) + (phase & 1) - 1;
        if (REFLECT) {
          ih = ih == -1 ? 1 : (ih == p.H ? p.H - 2 : ih);
          iw = iw == -1 ? 1 : (iw == p.W ? p.W - 2 : iw);
        }
      } else if constexpr (DG2) {   // dense dy, 3x3 tiles of the padded-gradient plane: rows 3 gy - 1 .. 3 gy + 2 (zeros outside)
        ih = 3 * gy - 1 + i; iw = 3 * gx - 1 + i;
      } else if constexpr (UPD) {   // source pixel 2 gy + 2 + (0, 1): plane rows one before .. two after, image row 2u + a
        ih = 2 * (2 * gy + 1 + i) + (phase >> 1); iw = 2 * (2 * gx + 1 + i) + (phase & 1);
      } else {
        ih = 2 * gy - 1 + i; iw = 2 * gx - 1 + i;
        if (REFLECT) {
          ih = ih < 0 ? -ih : (ih >= p.H ? 2 * p.H - 2 - ih : ih);
          iw = iw < 0 ? -iw : (iw >= p.W ? 2 * p.W - 2 - iw : iw);
        } else if (MODE == 1 && p.edge) {   // replicated edge (phase convolutions of the sub-pixel up-sampling layers)
          ih = min(max(ih, 0), p.H - 1); iw = min(max(iw, 0), p.W - 1);
        }
      }
      ro[i] = (unsigned)ih < (unsigned)p.H ? (gb * p.H + ih) * p.W * p.xc : -1;
      co[i] = (unsigned)iw < (unsigned)p.W ? iw * p.xc + 2 * cp : -1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) off[i * 4 + j] = (ro[i] >= 0 && co[j] >= 0) ? (unsigned)(ro[i] + co[j]) * 4u : 0x80000000u;
  };
  // Phase-major contractions (PHK): the 16 offsets change with the phase (row parity r = phase >> 1 moves the four row terms,
  // column parity the four column terms).  Re-deriving them inside the loop keeps the tile coordinates, p.H / p.W / p.xc and
  // the reflection logic live next to 128 accumulators + the prefetched U fragments -- over the 256-register limit (9 - 11
  // spilled registers in the 4x4 / stride 2 forward kernels).  Instead the prologue writes the thread's eight row terms (r = 0, 1)
  // and eight column terms (s = 0, 1) to a private LDS column behind the V buffers, and a phase change is 8 ds_read_b32 + 16
  // saturating adds: nothing but that column's address survives the prologue.
  unsigned* const otab = reinterpret_cast<unsigned*>(smem + 2 * VBUF) + (tid & 255);   // [16][256]: entry k of this thread at k * 256
  auto term_table = [&]() {
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int ih, iw;
        if constexpr (S2) {
          ih = 2 * (3 * gy + i) + par - 1;
          iw = 2 * (3 * gx + i) + par - 1;
          if (REFLECT) {
            ih = ih == -1 ? 1 : (ih == p.H ? p.H - 2 : ih);
            iw = iw == -1 ? 1 : (iw == p.W ? p.W - 2 : iw);
          }
        } else {   // UPD
          ih = 2 * (2 * gy + 1 + i) + par; iw = 2 * (2 * gx + 1 + i) + par;
        }
        // an invalid term carries bit 31: the sum of two terms saturates to 0x80000000 (beyond any buffer) when either is set
        otab[(par * 4 + i) * 256] = (unsigned)ih < (unsigned)p.H ? (unsigned)((gb * p.H + ih) * p.W * p.xc) * 4u : 0x80000000u;
        otab[(8 + par * 4 + i) * 256] = (unsigned)iw < (unsigned)p.W ? (unsigned)(iw * p.xc + 2 * cp) * 4u : 0x80000000u;
      }
  };
  auto load_off = [&](int phase) {
    unsigned ro[4], co[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ro[i] = otab[((phase >> 1) * 4 + i) * 256];
      co[i] = otab[(8 + (phase & 1) * 4 + i) * 256];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) off[i * 4 + j] = ((ro[i] | co[j]) & 0x80000000u) ? 0x80000000u : ro[i] + co[j];
  };
  if constexpr (PHK) {
    if (loader) { term_table(); load_off(0); }
  } else {
    make_off(0);
  }
  int off_phase = 0;
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  f32x2 d[16];
  auto load_raw = [&](int c) {
    int cc = c;
    if constexpr (PHK) {
      const int phase = c / p.cpp;
      cc = c - phase * p.cpp;
      if (phase != off_phase) { load_off(phase); off_phase = phase; }   // wave-uniform
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) d[q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xres, off[q], cc * (WK * 4), 0));
  };
  const int vpos = tl * 8 + ((cp ^ (((tl >> 3) & 1) << 1) ^ ((tl >> 4) & 3)) << 1);
  const bool e_top = gy == 0, e_bot = gy == p.th - 1, e_left = gx == 0, e_right = gx == p.tw - 1;
  auto transform_store = [&](int buf) {
    const f32x2 z = {0.f, 0.f};
    if constexpr (MODE == 2) {
      // (Skipping this arithmetic in blocks that hold no border tile -- a block-uniform branch -- measured 196 us against 182:
      // the branch costs the loader more than the 48 selects and adds it saves.  Kept unconditional.)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d[12 + j] += e_top ? d[4 + j] : z;
        d[0 + j] += e_bot ? d[8 + j] : z;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d[i * 4 + 3] += e_left ? d[i * 4 + 1] : z;
        d[i * 4 + 0] += e_right ? d[i * 4 + 2] : z;
      }
    }
    f32x2 u[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // B^T d
      u[0 + j] = d[0 + j] - d[8 + j];
      u[4 + j] = d[4 + j] + d[8 + j];
      u[8 + j] = d[8 + j] - d[4 + j];
      u[12 + j] = d[4 + j] - d[12 + j];
    }
    float* V = smem + buf * VBUF + vpos;
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // (.) B
      *reinterpret_cast<f32x2*>(V + (i * 4 + 0) * 512) = u[i * 4 + 0] - u[i * 4 + 2];
      *reinterpret_cast<f32x2*>(V + (i * 4 + 1) * 512) = u[i * 4 + 1] + u[i * 4 + 2];
      *reinterpret_cast<f32x2*>(V + (i * 4 + 2) * 512) = u[i * 4 + 2] - u[i * 4 + 1];
      *reinterpret_cast<f32x2*>(V + (i * 4 + 3) * 512) = u[i * 4 + 1] - u[i * 4 + 3];
    }
  };
  // U: the two frequency planes of a wave are private to it, so they never touch LDS: per chunk the wave reads its
  // 2 x 64 x 8 weights as four global_load_dwordx4 in fragment order (prepared image: [chunk][N block][f][half][lane][4] --
  // lane (n, kq) holds (k pair kq of channel n) for the 16-channel tiles 2*half and 2*half+1)
  const int phase = blockIdx.y;
  const f32x4* const ug = reinterpret_cast<const f32x4*>(p.u + phase * p.u_phase + (long long)n_blk * UBUF) + wave * 2 * 128 + lane;
  const long long u_chunk = (long long)p.NB * UBUF / 4;   // f32x4 per chunk
  struct UFrag { f32x4 v[2][2]; };   // [fq][half]
  auto load_u = [&](int c, UFrag& u) {
    const f32x4* g = ug + c * u_chunk;
#pragma unroll
    for (int fq = 0; fq < 2; ++fq)
#pragma unroll
      for (int h = 0; h < 2; ++h) u.v[fq][h] = g[fq * 128 + h * 64];
  };

  // ---- MFMA: wave owns frequencies 2*wave, 2*wave+1 ----
  f32x4 acc[2][4][4];
#pragma unroll
  for (int fq = 0; fq < 2; ++fq)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[fq][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment of row r = lane & 15 of 16-row tile mt, k pair kq = lane >> 4: slot kq ^ (2 * ((r >> 3) & 1)) ^ mt.  A
  // half-wave's ds_read_b64 (banks mod 64) then covers 16 rows x 2 slots = 64 distinct banks; the mt term keeps the four
  // reads of a fragment set from being a constant apart, or the compiler fuses them into ds_read2st64_b64, which banks
  // mod 32 in 16-lane groups (2-way conflicts here: measured 36 % of the LDS cycles) and moves half the bytes per clock.
  // The eight fragment positions (2 frequencies x 4 row tiles) are formed ONCE and made opaque to the compiler: it then
  // cannot see that the two frequencies of a row tile are a constant apart (it would fuse the pair into ds_read2st64_b64, see
  // above), and the loop carries no address arithmetic (the buffer offset is an immediate of the read).  Hiding the address
  // inside the loop instead cost a v_mov + v_lshl per read: 16 vector instructions per wave and chunk, and on this chip
  // vector instructions of either wave of a SIMD are not hidden behind v_mfma_f32_16x16x4_f32 (tools/ubench/mfma_coissue.hip).
  // (MODE 2 -- the border fold keeps 16 more values live in the loader -- has no registers for the four extra positions: the
  // hoisted form made the compiler shuffle registers around the transform and measured 6 % slower there, so that variant hides
  // the address per read as before.)
  constexpr bool HOIST = MODE != 2;
  int fpos[2][4];
#pragma unroll
  for (int fq = 0; fq < 2; ++fq)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      fpos[fq][mt] = (HOIST ? (wave * 2 + fq) * 512 : 0) + mt * 128 + (lane & 15) * 8 + (((lane >> 4) ^ (((lane >> 3) & 1) << 1) ^ mt) << 1);
      if constexpr (HOIST) asm("" : "+v"(fpos[fq][mt]));
    }
  auto compute = [&](int buf, const UFrag& u) {
    const float* Vf = smem + buf * VBUF;
    f32x2 a[2][4];
#pragma unroll
    for (int fq = 0; fq < 2; ++fq)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if constexpr (HOIST) {
          a[fq][mt] = *reinterpret_cast<const f32x2*>(Vf + fpos[fq][mt]);
        } else {
          int o = (wave * 2 + fq) * 512 + fpos[0][mt];
          asm("" : "+v"(o));   // opaque to the compiler: every fragment read stays a ds_read_b64 of its own (see above)
          a[fq][mt] = *reinterpret_cast<const f32x2*>(Vf + o);
        }
      }
#pragma unroll
    for (int fq = 0; fq < 2; ++fq)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[fq][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[fq][mt][t], u.v[fq][nt >> 1][(nt & 1) * 2 + t], acc[fq][mt][nt], 0, 0, 0);
  };

  // One barrier per chunk publishes V(c+1) and frees V(c-1)'s buffer; it is the only synchronisation left.
  // Every wave requests U(c+1) while chunk c multiplies (two register sets, ping-pong): inside the step the 4 MB image of a
  // layer is not L2-resident as it is in a back-to-back timing loop, and a request issued only at the start of its own chunk
  // (tried for the loader waves, which are short of registers) made the step 1.4 % slower while the loop timing improved.
  const int nc = p.K / WK;
#ifdef WINO_STAMP
  long long st_acc[4] = {0, 0, 0, 0};
  long long st_last = __builtin_amdgcn_s_memtime();
#endif
  if (loader) {
    UFrag u0, u1;
    load_raw(0);
    load_u(0, u0);
    transform_store(0);
    if (nc > 1) load_raw(1);
    __syncthreads();
    WSTAMP(3);
    auto body = [&](int c, int cur, const UFrag& u, UFrag& un) {
      if (c + 1 < nc) {
        load_u(c + 1, un);
        transform_store(cur ^ 1);
        if (c + 2 < nc) load_raw(c + 2);
      }
      WSTAMP(0);
      compute(cur, u);
      WSTAMP(1);
      __syncthreads();
      WSTAMP(2);
    };
    for (int c = 0; c < nc; c += 2) {
      body(c, 0, u0, u1);
      if (c + 1 < nc) body(c + 1, 1, u1, u0);
    }
  } else {
    UFrag u0, u1;
    load_u(0, u0);
    __syncthreads();
    WSTAMP(3);
    for (int c = 0; c < nc; c += 2) {
      if (c + 1 < nc) load_u(c + 1, u1);
      WSTAMP(0);
      compute(0, u0);
      WSTAMP(1);
      __syncthreads();
      WSTAMP(2);
      if (c + 1 < nc) {
        if (c + 2 < nc) load_u(c + 2, u0);
        WSTAMP(0);
        compute(1, u1);
        WSTAMP(1);
        __syncthreads();
        WSTAMP(2);
      }
    }
  }
#ifdef WINO_STAMP
  if (blockIdx.x < 8 && blockIdx.y == 0 && lane == 0)
    for (int i = 0; i < 4; ++i) g_wino_stamps[(blockIdx.x * 8 + wave) * 4 + i] = st_acc[i];
#endif
  // ---- epilogue: M[f][tile][32 channels] planes through LDS, two halves ----
  // every per-thread index of the epilogue derives from this opaque copy, so none of them can be formed before the main loop and
  // parked in scratch memory across it (the loop runs at the 256-register limit)
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63;
  // ... and the tile-list divisions of the epilogue use opaque copies of their divisors: otherwise the compiler shares the
  // reciprocals it formed for the loader's divisions in the prologue and carries them through the loop in (spilled) registers
  int tw_e = p.tw, th_e = p.th;
  asm volatile("" : "+s"(tw_e), "+s"(th_e));
  const int co = tid_e & 31;
  const float slope = p.slope;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int fq = 0; fq < 2; ++fq)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ntl = 0; ntl < 2; ++ntl)
#pragma unroll
          for (int r = 0; r < 4; ++r)   // C/D map: col = lane & 15, row = 4 * (lane >> 4) + r
            smem[((wave * 2 + fq) * 64 + mt * 16 + 4 * (lane_e >> 4) + r) * MLD + ntl * 16 + (lane_e & 15)] = acc[fq][mt][half * 2 + ntl][r];
    __syncthreads();
    const int n = n_blk * 64 + half * 32 + co;
    const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int t2 = (tid_e >> 5) + 16 * it;
      float m[16];
#pragma unroll
      for (int f = 0; f < 16; ++f) m[f] = smem[(f * 64 + t2) * MLD + co];
      if constexpr (DG2) {
        float s[12];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] = (m[0 + j] + m[4 + j]) + m[8 + j];
          s[4 + j] = m[4 + j] - m[8 + j];
          s[8 + j] = (m[4 + j] + m[8 + j]) - m[12 + j];
        }
        const int t = m_blk0 * 64 + t2;
        if (t < p.B * th_e * tw_e) {
          const int tx = t % tw_e, ty = (t / tw_e) % th_e, tb = t / (tw_e * th_e);
          const int ph = phase >> 1, pw = phase & 1;
          float v[3][3];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            v[i][0] = (s[i * 4 + 0] + s[i * 4 + 1]) + s[i * 4 + 2];
            v[i][1] = s[i * 4 + 1] - s[i * 4 + 2];
            v[i][2] = (s[i * 4 + 1] + s[i * 4 + 2]) - s[i * 4 + 3];
          }
          // plane index u = 3 ty + i  <->  padded row 2u + ph  <->  image row 2u + ph - 1.  Fold rows: u = 0 of parity 0 onto
          // u = 1; u = H (the plane's last) of parity 1 onto u = H - 1; columns likewise.  p.H, p.W = extent of dy.
          const int fr = ph == 0 ? 0 : p.H, fc = pw == 0 ? 0 : p.W;   // folding plane row / column; its target sits one step inside
          if (REFLECT) {
            // every index below is a compile-time constant: a run-time `v[i + dr][j]` sends the 3x3 values to scratch memory
            const int ir = fr - 3 * ty, jc = fc - 3 * tx;   // position of the folding row / column inside this tile (if 0..2)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              if (ph == 0) { if (ir == 0) v[1][j] += v[0][j]; }
              else { if (ir == 1) v[0][j] += v[1][j]; else if (ir == 2) v[1][j] += v[2][j]; }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              if (pw == 0) { if (jc == 0) v[i][1] += v[i][0]; }
              else { if (jc == 1) v[i][0] += v[i][1]; else if (jc == 2) v[i][1] += v[i][2]; }
            }
          }
          float* yb = p.y + tb * p.y_sb + n;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int u = 3 * ty + i, ir = 2 * u + ph - 1;
            if (u == fr || ir < 0 || ir >= p.Ho) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const int w = 3 * tx + j, ic = 2 * w + pw - 1;
              if (w == fc || ic < 0 || ic >= p.Wo) continue;
              yb[(long long)ir * p.y_sh + (long long)ic * p.y_sw] = v[i][j];
            }
          }
        }
      } else if constexpr (S2) {   // A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 -1]: 3x3 pixels, the ragged last tile clipped
        float s[12];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] = (m[0 + j] + m[4 + j]) + m[8 + j];
          s[4 + j] = m[4 + j] - m[8 + j];
          s[8 + j] = (m[4 + j] + m[8 + j]) - m[12 + j];
        }
        const int t = m_blk0 * 64 + t2;
        if (t < p.B * th_e * tw_e) {
          const int tx = t % tw_e, ty = (t / tw_e) % th_e, tb = t / (tw_e * th_e);
          float* yp = p.y + tb * p.y_sb + (long long)(3 * ty) * p.y_sh + (long long)(3 * tx) * p.y_sw + n;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const float v[3] = {(s[i * 4 + 0] + s[i * 4 + 1]) + s[i * 4 + 2], s[i * 4 + 1] - s[i * 4 + 2],
                                (s[i * 4 + 1] + s[i * 4 + 2]) - s[i * 4 + 3]};
#pragma unroll
            for (int j = 0; j < 3; ++j)
              if (3 * ty + i < p.Ho && 3 * tx + j < p.Wo) yp[i * p.y_sh + j * p.y_sw] = wino_act(v[j] + bv, p.act, slope);
          }
        }
      } else {
        const int ty = bty * 8 + (t2 >> 3), tx = btx * 8 + (t2 & 7);
        float s[8];   // A^T m: rows 0, 1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] = (m[0 + j] + m[4 + j]) + m[8 + j];
          s[4 + j] = (m[4 + j] - m[8 + j]) - m[12 + j];
        }
        if (ty < p.th && tx < p.tw) {
          const long long yo = (phase >> 1) * p.y_prow + (phase & 1) * p.y_pcol + b * p.y_sb + (long long)(2 * ty) * p.y_sh +
                               (long long)(2 * tx) * p.y_sw + n;
          float* yp = p.y + yo;
          float v0 = ((s[0] + s[1]) + s[2]) + bv, v1 = ((s[1] - s[2]) - s[3]) + bv;
          float v2 = ((s[4] + s[5]) + s[6]) + bv, v3 = ((s[5] - s[6]) - s[7]) + bv;
          if (p.add != nullptr) {   // backward-data of the first conv of a ResBlock: + the gradient of the skip connection
            const float* ap = p.add + yo;
            v0 += ap[0]; v1 += ap[p.y_sw]; v2 += ap[p.y_sh]; v3 += ap[p.y_sh + p.y_sw];
          }
          yp[0] = wino_act(v0, p.act, slope);
          yp[p.y_sw] = wino_act(v1, p.act, slope);
          yp[p.y_sh] = wino_act(v2, p.act, slope);
          yp[p.y_sh + p.y_sw] = wino_act(v3, p.act, slope);
        }
      }
    }
    if (half == 0) __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Backward-weight.  S[f][co][ci] = sum over tiles of E[f][tile][co] * V[f][tile][ci] with E = A dY A^T (the 2x2 tile of
// dy spread to 4x4) and V = B^T d B (the forward's input transform): sixteen GEMMs contracting over the tiles, 16
// products per tile and channel pair instead of 36; dw = G^T S G afterwards.  One block = 64 output x 64 input channels x
// 16 frequencies over a slice of the tile range (split-K: the slices' partial S go to slabs, wino_wgrad_reduce_kernel
// adds them and applies G).  Chunk = 8 tiles: wave w loads tile w of the chunk, lane = channel (256-byte rows of x and
// dy, all addressing on the scalar unit), transforms in registers and writes E / V as [f][tile pair][channel][2] so that
// the MFMA fragments (channel rows, 4 tile pairs deep) are conflict-free ds_read_b64.  Same accumulator layout, double
// buffering and one barrier per chunk as the forward kernel; all eight waves load and transform here (waves 0-3 before their
// MFMAs, 4-7 after): giving the loads to four waves of two tiles each, as the forward does, was slower (275 vs 222 us).
// ------------------------------------------------------------------------------------------------------------------
constexpr int WG_PLANE = 4 * 64 * 2;        // floats per frequency plane: [tile pair 4][channel 64][2]
constexpr int WG_BUF = 16 * WG_PLANE;       // floats per operand buffer (32 KiB)

// S2: the 4x4 / stride 2 layers, F(3x3, 2x2): launch phase = parity (r, s) of the filter tap; its 2x2 gradient contracts the
// 3x3 tiles of dy (A dY A^T with A = [1 0 0; 1 1 1; 1 -1 1; 0 0 -1]) against the 4x4 patches of input phase (r, s).
//
// Round 3: a wave now loads a PAIR of tiles of ONE operand -- waves 0-3 the two input patches of tile pair (wave & 3), waves
// 4-7 the two dy tiles -- and stores both tiles of a frequency with one ds_write_b64 ([f][tile pair][channel][2] is exactly
// that pair).  Before, every wave loaded one tile of BOTH operands and stored 32 single dwords; per SIMD and chunk the two
// resident waves issued ~300 vector / LDS / memory instructions next to their 128 MFMAs, now ~190 -- and on this chip those are
// not hidden behind v_mfma_f32_16x16x4_f32 of the partner wave (tools/ubench/mfma_coissue.hip: additive), so they are what the
// kernel's distance from the matrix roofline consists of.  FAST (host-checked: every tile of every chunk exists and every patch
// position is inside the image after reflection) drops the per-load validity branches and the zero fills.
// RING: phase gradients of an up-sampling layer (p.reflect == 2, p.ring_mask): replicated source edge, the outermost ring of dy
// read as 0.  A template argument, not a run-time branch: the uniform tests alone cost the 3x3 trunk layers 6 % (208 -> 221 us).
template <bool S2, bool FAST, bool RING = false>
__global__ __launch_bounds__(512) void conv_wino_wgrad_kernel(WinoWgradParams p) {
  __shared__ __attribute__((aligned(16))) float smem[4 * WG_BUF];   // E0 E1 V0 V1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order (blocks b and b + 8 share an XCD and its L2): every XCD gets one contiguous run of block indices, so the
  // CB x NB blocks of a tile-range split -- which read the same x tiles (NB times) and the same dy tiles (CB times) -- meet in
  // one L2 instead of fetching them through eight
  int blk = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blk & 7, idx = blk >> 3;
    blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cib = blk % p.CB; blk /= p.CB;
  const int cob = blk % p.NB;
  const int split = blk / p.NB;
  const int phase = blockIdx.y;
  const int c_begin = split * p.cps;
  const int total_chunks = (p.tiles + 7) >> 3;
  const int nc = min(p.cps, total_chunks - c_begin);

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
  const unsigned xlane = (unsigned)(cib * 64 + lane) * 4u, ylane = (unsigned)(cob * 64 + lane) * 4u;
  const bool xside = wave < 4;            // waves 0-3: input patches -> V;  waves 4-7: output-gradient tiles -> E
  const int pair = wave & 3;              // tile pair of the chunk this wave loads (tiles 2 pair, 2 pair + 1)
  const bool want_db_blk = p.db_part != nullptr && cib == 0 && (!S2 || phase == 0);   // block-uniform
  const bool want_db = want_db_blk && !xside;   // the dy-side waves hold the tiles' sums

  constexpr int NG = S2 ? 9 : 4;   // dy values per tile
  // element e of every pair register = tile 2 pair + e: the loads land in the halves of a register pair, the transforms are
  // packed adds over the pair and a frequency's two tiles leave as one ds_write_b64 -- no register shuffling in between
  f32x2 d[16], g[NG];
  float dbs = 0.f;
  // tiles 2 pair, 2 pair + 1 of chunk c: everything but the lane's channel offset is wave-uniform (scalar unit).  The tile
  // coordinates are carried from chunk to chunk (load_raw is called for c = 0, 1, 2, ... in order: + 8 tiles each time)
  // instead of being divided out of the tile index every time.
  int cur_t = c_begin * 8 + 2 * pair, cur_tx, cur_ty, cur_b;
  {
    const int r = cur_t / p.tw;
    cur_tx = cur_t - r * p.tw; cur_b = r / p.th; cur_ty = r - cur_b * p.th;
  }
  auto load_raw = [&](int c) {
    (void)c;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int t = cur_t + e;
      if (FAST || (S2 && RING) || t < p.tiles) {
        int tx = cur_tx + e, ty = cur_ty, b = cur_b;
        if (tx >= p.tw) { tx -= p.tw; if (++ty >= p.th) { ty = 0; ++b; } }
        if (xside) {
          int ro[4], co[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int ih, iw;
            if constexpr (S2) {   // row q of input phase r = image row 2q + r - 1; the padded rows -1 and H reflect (or read 0)
              ih = 2 * (3 * ty + i) + (phase >> 1) - 1; iw = 2 * (3 * tx + i) + (phase & 1) - 1;
              if constexpr (RING) {   // S2 + RING = reflect-padded layer whose 3x3 tiles overhang the output (XCLAMP, see the launch)
                ih = ih == -1 ? 1 : (ih == p.H ? p.H - 2 : min(ih, p.H - 1));
                iw = iw == -1 ? 1 : (iw == p.W ? p.W - 2 : min(iw, p.W - 1));
              } else if (p.reflect) {
                ih = ih == -1 ? 1 : (ih == p.H ? p.H - 2 : ih);
                iw = iw == -1 ? 1 : (iw == p.W ? p.W - 2 : iw);
              }
            } else {
              ih = 2 * ty + p.xo + i; iw = 2 * tx + p.xo + i;
              if constexpr (RING) {   // replicated edge
                ih = min(max(ih, 0), p.H - 1); iw = min(max(iw, 0), p.W - 1);
              } else if (p.reflect) {
                ih = ih < 0 ? -ih : (ih >= p.H ? 2 * p.H - 2 - ih : ih);
                iw = iw < 0 ? -iw : (iw >= p.W ? 2 * p.W - 2 - iw : iw);
              }
            }
            if constexpr (FAST || (S2 && RING)) {
              ro[i] = (b * p.H + ih) * p.W; co[i] = iw;
            } else {
              ro[i] = (unsigned)ih < (unsigned)p.H ? (b * p.H + ih) * p.W : -1;
              co[i] = (unsigned)iw < (unsigned)p.W ? iw : -1;
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              // the 4x4 / stride 2 layers mark an invalid position by a lane offset beyond the buffer (the load returns 0) instead
              // of branching around the load: 254 -> 245 us; for the sub-pixel layers the branch form measured 1 % faster
              if constexpr (S2) {
                const bool ok = FAST || RING || (ro[i] >= 0 && co[j] >= 0);
                d[i * 4 + j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    xres, ok ? xlane : 0x80000000u, ok ? (ro[i] + co[j]) * p.Cin * 4 : 0, 0));
              } else {
                if (FAST || (ro[i] >= 0 && co[j] >= 0))
                  d[i * 4 + j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, xlane, (ro[i] + co[j]) * p.Cin * 4, 0));
                else
                  d[i * 4 + j][e] = 0.f;
              }
            }
        } else {
          const int sw = (int)p.dy_sw * 4, sh = (int)p.dy_sh * 4;
          if constexpr (S2) {
            const int y0 = (int)(p.dy_off + b * p.dy_sb) * 4 + 3 * ty * sh + 3 * tx * sw;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < 3; ++j) {
                const bool ok = FAST || (3 * ty + i < p.Ho && 3 * tx + j < p.Wo);
                g[i * 3 + j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    yres, ok ? ylane : 0x80000000u, ok ? y0 + i * sh + j * sw : 0, 0));
              }
          } else {
            const int y0 = (int)(p.dy_off + (phase >> 1) * p.dy_prow + (phase & 1) * p.dy_pcol + b * p.dy_sb) * 4 + 2 * ty * sh + 2 * tx * sw;
            if constexpr (RING) {   // the outermost ring of the up-sampled output belongs to the frame launch: read it as 0
              const bool r0 = !((phase >> 1) == 0 && ty == 0), r1 = !((phase >> 1) == 1 && ty == p.th - 1);
              const bool c0 = !((phase & 1) == 0 && tx == 0), c1 = !((phase & 1) == 1 && tx == p.tw - 1);
              g[0][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yres, r0 && c0 ? ylane : 0x80000000u, y0, 0));
              g[1][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yres, r0 && c1 ? ylane : 0x80000000u, y0 + sw, 0));
              g[2][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yres, r1 && c0 ? ylane : 0x80000000u, y0 + sh, 0));
              g[3][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yres, r1 && c1 ? ylane : 0x80000000u, y0 + sh + sw, 0));
            } else {
              g[0][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yres, ylane, y0, 0));
              g[1][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yres, ylane, y0 + sw, 0));
              g[2][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yres, ylane, y0 + sh, 0));
              g[3][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yres, ylane, y0 + sh + sw, 0));
            }
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q][e] = 0.f;
#pragma unroll
        for (int q = 0; q < NG; ++q) g[q][e] = 0.f;
      }
    }
    cur_t += 8; cur_tx += 8;
    while (cur_tx >= p.tw) { cur_tx -= p.tw; if (++cur_ty >= p.th) { cur_ty = 0; ++cur_b; } }
  };
  // plane position of (tile pair `pair`, channel lane): one b64 = the pair's two tiles; odd pairs swap the two 16-channel
  // halves of every 32 so that the half-wave fragment reads below (pairs {0,1} or {2,3}) cover all 64 banks
  const int wpos = (pair * 64 + (lane ^ ((pair & 1) << 4))) * 2;
  auto transform_store = [&](int buf) {
    if (xside) {
      float* V = smem + (2 + buf) * WG_BUF + wpos;
      f32x2 u[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // B^T d
        u[0 + j] = d[0 + j] - d[8 + j];
        u[4 + j] = d[4 + j] + d[8 + j];
        u[8 + j] = d[8 + j] - d[4 + j];
        u[12 + j] = d[4 + j] - d[12 + j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // (.) B
        *reinterpret_cast<f32x2*>(V + (i * 4 + 0) * WG_PLANE) = u[i * 4 + 0] - u[i * 4 + 2];
        *reinterpret_cast<f32x2*>(V + (i * 4 + 1) * WG_PLANE) = u[i * 4 + 1] + u[i * 4 + 2];
        *reinterpret_cast<f32x2*>(V + (i * 4 + 2) * WG_PLANE) = u[i * 4 + 2] - u[i * 4 + 1];
        *reinterpret_cast<f32x2*>(V + (i * 4 + 3) * WG_PLANE) = u[i * 4 + 1] - u[i * 4 + 3];
      }
    } else {
      float* E = smem + buf * WG_BUF + wpos;
      if constexpr (S2) {
        if (want_db) {
          const f32x2 sum = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7])) + g[8];
          dbs += sum[0];
          dbs += sum[1];
        }
        // E = A dY A^T, A = [1 0 0; 1 1 1; 1 -1 1; 0 0 -1]
        f32x2 r[4][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          r[0][j] = g[j];
          r[1][j] = (g[j] + g[6 + j]) + g[3 + j];
          r[2][j] = (g[j] + g[6 + j]) - g[3 + j];
          r[3][j] = -g[6 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          *reinterpret_cast<f32x2*>(E + (i * 4 + 0) * WG_PLANE) = r[i][0];
          *reinterpret_cast<f32x2*>(E + (i * 4 + 1) * WG_PLANE) = (r[i][0] + r[i][2]) + r[i][1];
          *reinterpret_cast<f32x2*>(E + (i * 4 + 2) * WG_PLANE) = (r[i][0] + r[i][2]) - r[i][1];
          *reinterpret_cast<f32x2*>(E + (i * 4 + 3) * WG_PLANE) = -r[i][2];
        }
      } else {
        if (want_db) {
          const f32x2 sum = (g[0] + g[1]) + (g[2] + g[3]);
          dbs += sum[0];
          dbs += sum[1];
        }
        // E = A dY A^T, A = [1 0; 1 1; 1 -1; 0 -1]
        const f32x2 r[4][2] = {{g[0], g[1]}, {g[0] + g[2], g[1] + g[3]}, {g[0] - g[2], g[1] - g[3]}, {-g[2], -g[3]}};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          *reinterpret_cast<f32x2*>(E + (i * 4 + 0) * WG_PLANE) = r[i][0];
          *reinterpret_cast<f32x2*>(E + (i * 4 + 1) * WG_PLANE) = r[i][0] + r[i][1];
          *reinterpret_cast<f32x2*>(E + (i * 4 + 2) * WG_PLANE) = r[i][0] - r[i][1];
          *reinterpret_cast<f32x2*>(E + (i * 4 + 3) * WG_PLANE) = -r[i][1];
        }
      }
    }
  };

  f32x4 acc[2][4][4];
#pragma unroll
  for (int fq = 0; fq < 2; ++fq)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[fq][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment: channel row r = lane & 15 of 16-channel tile mt, tile pair kq = lane >> 4.  Positions formed once and made
  // opaque (see the forward kernel): no address arithmetic in the loop, and the reads of the two frequencies are not fused
  int fpos[2][4];
#pragma unroll
  for (int fq = 0; fq < 2; ++fq)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      fpos[fq][mt] = (wave * 2 + fq) * WG_PLANE + ((lane >> 4) * 64 + ((mt * 16 + (lane & 15)) ^ (((lane >> 4) & 1) << 4))) * 2;
      asm("" : "+v"(fpos[fq][mt]));
    }
  auto compute = [&](int buf, int fq) {
    const float* Ef = smem + buf * WG_BUF;
    const float* Vf = smem + (2 + buf) * WG_BUF;
    f32x2 a[4], bb[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bb[nt] = *reinterpret_cast<const f32x2*>(Vf + fpos[fq][nt]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const f32x2*>(Ef + fpos[fq][mt]);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[fq][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][t], bb[nt][t], acc[fq][mt][nt], 0, 0, 0);
  };

  if (nc > 0) {
    load_raw(0);
    transform_store(0);
    if (nc > 1) load_raw(1);
  }
  __syncthreads();
  for (int c = 0; c < nc; ++c) {
    const int cur = c & 1;
    const bool more = c + 1 < nc;
    // input-side waves transform before their MFMAs, dy-side waves after (all before: +4 %, swapped: +2 %)
    if (more && wave < 4) {
      transform_store(cur ^ 1);
      if (c + 2 < nc) load_raw(c + 2);
    }
    compute(cur, 0);
    compute(cur, 1);
    if (more && wave >= 4) {
      transform_store(cur ^ 1);
      if (c + 2 < nc) load_raw(c + 2);
    }
    __syncthreads();
  }

  // ---- partial S -> slab[split][f][co][ci] (C/D map: row = co = 4 * (lane >> 4) + r, col = ci = lane & 15) ----
  float* const sl = p.slab + ((long long)phase * p.ksplit + split) * 16 * p.Cout * p.Cin;
#pragma unroll
  for (int fq = 0; fq < 2; ++fq)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sl[((long long)(wave * 2 + fq) * p.Cout + cob * 64 + mt * 16 + 4 * (lane >> 4) + r) * p.Cin + cib * 64 + nt * 16 + (lane & 15)] =
              acc[fq][mt][nt][r];
  if (want_db_blk) {   // bias: the dy-side waves hold the sums of their tiles for channel `lane` (the others 0)
    smem[wave * 64 + lane] = dbs;
    __syncthreads();
    if (wave == 0) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += smem[w * 64 + lane];
      p.db_part[(phase * p.ksplit + split) * p.Cout + cob * 64 + lane] = s;
    }
  }
}

// dw[co][r][s][ci] = beta * dw + (G^T S G)[r][s],  S = sum over splits; thread = (co, ci), ci fastest.  The last blocks add
// up the bias partials.
__global__ void wino_wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ db_part, float* __restrict__ dw,
                                         long long dw_phase, float* __restrict__ db, int Cout, int Cin, int ksplit, float beta,
                                         float beta_b, int pair_blocks, int s2) {
  if ((int)blockIdx.x >= pair_blocks) {   // bias: all phases and splits (s2: the partials of phase 0 only), phase-0 blocks only
    const int co = (blockIdx.x - pair_blocks) * blockDim.x + threadIdx.x;
    if (co < Cout && blockIdx.y == 0) {
      float s = 0.f;
      for (int k = 0; k < ksplit * (s2 ? 1 : (int)gridDim.y); ++k) s += db_part[k * Cout + co];
      db[co] = beta_b == 0.f ? s : beta_b * db[co] + s;
    }
    return;
  }
  slab += (long long)blockIdx.y * ksplit * 16 * Cout * Cin;
  dw += blockIdx.y * dw_phase;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Cout * Cin) return;
  const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
  const long long plane = (long long)Cout * Cin;
  float S[16];
#pragma unroll
  for (int f = 0; f < 16; ++f) S[f] = 0.f;
  for (int k = 0; k < ksplit; ++k) {
    const float* sp = slab + (long long)k * 16 * plane + idx;
#pragma unroll
    for (int f = 0; f < 16; ++f) S[f] += sp[f * plane];
  }
  if (s2) {   // 2x2 gradient of filter-tap parity (r, s) = launch phase: G^T = [1 .5 .5 0; 0 .5 -.5 1]; dw is [Cout][4][4][Cin]
    const int r = blockIdx.y >> 1, sf = blockIdx.y & 1;
    float t[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = S[j] + 0.5f * (S[4 + j] + S[8 + j]);
      t[1][j] = 0.5f * (S[4 + j] - S[8 + j]) + S[12 + j];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float v[2] = {t[a][0] + 0.5f * (t[a][1] + t[a][2]), 0.5f * (t[a][1] - t[a][2]) + t[a][3]};
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float* o = dw - blockIdx.y * dw_phase + (((long long)co * 4 + 2 * a + r) * 4 + 2 * b + sf) * Cin + ci;
        *o = beta == 0.f ? v[b] : beta * *o + v[b];
      }
    }
    return;
  }
  // G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
  float t[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float h = 0.5f * (S[4 + j] + S[8 + j]), m = 0.5f * (S[4 + j] - S[8 + j]);
    t[0][j] = S[j] + h;
    t[1][j] = m;
    t[2][j] = h + S[12 + j];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float h = 0.5f * (t[r][1] + t[r][2]), m = 0.5f * (t[r][1] - t[r][2]);
    const float v[3] = {t[r][0] + h, m, h + t[r][3]};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float* o = dw + (((long long)co * 3 + r) * 3 + s) * Cin + ci;
      *o = beta == 0.f ? v[s] : beta * *o + v[s];
    }
  }
}

}  // namespace

#ifdef WINO_STAMP
extern "C" int munit_debug_wino_stamps(long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino_stamps), sizeof(long long) * 8 * 8 * 4) == hipSuccess ? 0 : -1;
}
#endif

bool munit_wino_wgrad_ok(int B, int H, int W, int Cin, int Cout) {
  if (MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINOGRAD") || MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINOGRAD_WGRAD")) return false;
  return Cin % 64 == 0 && Cout % 64 == 0 && H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2 &&
         (long long)B * H * W * std::max(Cin, Cout) < (1ll << 29);
}

int munit_wino_wgrad_splits(long long tiles, int Cin, int Cout, int phases) {
  const int chunks = cdiv(tiles, 8);
  const int pairs = (Cin / 64) * (Cout / 64) * std::max(1, phases);
  int ksplit = std::max(1, std::min(chunks, cdiv(256, pairs)));   // one block per CU where the tile range allows
  const int cps = cdiv(chunks, ksplit);
  return cdiv(chunks, cps);
}

size_t munit_wino_wgrad_workspace(long long tiles, int Cin, int Cout, int phases) {
  const size_t k = (size_t)munit_wino_wgrad_splits(tiles, Cin, Cout, phases) * std::max(1, phases);
  return align_up(k * 16 * Cin * Cout * sizeof(float), 256) + align_up(k * Cout * sizeof(float), 256);
}

int munit_wino_wgrad_launch(WinoWgradParams p, float* dw, long long dw_phase, float* db, float beta, float beta_b, void* ws,
                            hipStream_t st) {
  p.phases = std::max(1, p.phases);
  p.tiles = p.B * p.th * p.tw;
  p.ksplit = munit_wino_wgrad_splits(p.tiles, p.Cin, p.Cout, p.phases);
  p.cps = cdiv(cdiv(p.tiles, 8), p.ksplit);
  p.CB = p.Cin / 64; p.NB = p.Cout / 64;
  p.slab = reinterpret_cast<float*>(ws);
  float* db_part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) +
                                            align_up((size_t)p.ksplit * p.phases * 16 * p.Cin * p.Cout * sizeof(float), 256));
  p.db_part = db != nullptr ? db_part : nullptr;
  const dim3 grid((unsigned)(p.CB * p.NB * p.ksplit), (unsigned)p.phases);
  // FAST: every tile of every chunk exists and every patch / dy position lies inside its tensor (after reflection)
  bool fast = p.tiles % 8 == 0 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINO_WGRAD_FAST");
  if (p.s2) fast = fast && p.reflect && p.H % 6 == 0 && p.W % 6 == 0 && p.Ho % 3 == 0 && p.Wo % 3 == 0;
  else fast = fast && (p.xo == -1 ? (p.reflect && p.H >= 2 && p.W >= 2 && 2 * p.th <= p.H && 2 * p.tw <= p.W)
                                  : (p.xo >= 0 && 2 * (p.th - 1) + p.xo + 3 < p.H && 2 * (p.tw - 1) + p.xo + 3 < p.W));
  if (p.s2) {
    // XCLAMP (<true, false, true>): a reflect-padded layer whose 3x3 tiles overhang the output (Ho % 3 != 0 -- every power-of-two
    // extent) with every tile of every chunk present.  The overhanging dy positions are read as 0 (guarded, 9 loads per tile), so
    // the input positions that meet ONLY them may hold any finite value: the x side -- 32 loads per wave and chunk -- clamps them
    // into the image and drops its per-load validity selects.  Exact in the sense of the transform (linear; the products with a
    // zero gradient cancel), and the op tests hold it to the same bound as the guarded form.
    const bool xclamp = !fast && p.reflect == 1 && p.tiles % 8 == 0 && p.H >= 2 && p.W >= 2 && !MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINO_WGRAD_FAST");
    if (fast) hipLaunchKernelGGL((conv_wino_wgrad_kernel<true, true>), grid, dim3(512), 0, st, p);
    else if (xclamp) hipLaunchKernelGGL((conv_wino_wgrad_kernel<true, false, true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_wino_wgrad_kernel<true, false>), grid, dim3(512), 0, st, p);
  } else if (p.ring_mask) {
    MUNIT_CHECK_ARG(p.reflect == 2 && p.xo == -1 && p.phases == 4, "conv_wino_wgrad: ring_mask goes with the replicated edge and 4 phases");
    if (fast) hipLaunchKernelGGL((conv_wino_wgrad_kernel<false, true, true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_wino_wgrad_kernel<false, false, true>), grid, dim3(512), 0, st, p);
  } else {
    MUNIT_CHECK_ARG(p.reflect != 2, "conv_wino_wgrad: the replicated edge goes with ring_mask");
    if (fast) hipLaunchKernelGGL((conv_wino_wgrad_kernel<false, true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_wino_wgrad_kernel<false, false>), grid, dim3(512), 0, st, p);
  }
  MUNIT_CHECK_LAUNCH("conv_wino_wgrad");
  const int pair_blocks = cdiv((long long)p.Cout * p.Cin, 256);
  const int bias_blocks = db != nullptr ? cdiv(p.Cout, 256) : 0;
  hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)(pair_blocks + bias_blocks), (unsigned)p.phases), dim3(256), 0, st, p.slab,
                     db_part, dw, dw_phase, db, p.Cout, p.Cin, p.ksplit, beta, beta_b, pair_blocks, p.s2);
  MUNIT_CHECK_LAUNCH("wino_wgrad_reduce");
  return MUNIT_OK;
}

bool munit_wino_ok(int B, int H, int W, int K, int N) {
  if (MUNIT_ENV_FLAG("MUNIT_DEBUG_NO_WINOGRAD")) return false;
  return K % 8 == 0 && N % 64 == 0 && H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2 &&
         (long long)B * H * W * K < (1ll << 29) && (long long)B * H * W * N < (1ll << 40);
}

int munit_wino_launch(const WinoParams& p, hipStream_t st) {
  const bool lin = p.s2 == 1 || p.s2 == 2;
  const long long blocks = lin ? (long long)cdiv((long long)p.B * p.th * p.tw, 64) * p.NB : (long long)p.B * p.bth * p.btw * p.NB;
  MUNIT_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "conv_wino: bad grid");
  const dim3 grid((unsigned)blocks, (unsigned)std::max(1, p.phases));
  if (p.s2 == 1) {
    if (p.mode == 0) hipLaunchKernelGGL((conv_wino_kernel<0, 1>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_wino_kernel<1, 1>), grid, dim3(512), 0, st, p);
    MUNIT_CHECK_LAUNCH("conv_wino_s2");
    return MUNIT_OK;
  }
  if (p.s2 == 3) {
    hipLaunchKernelGGL((conv_wino_kernel<1, 3>), grid, dim3(512), 0, st, p);
    MUNIT_CHECK_LAUNCH("conv_wino_up_dgrad");
    return MUNIT_OK;
  }
  if (p.s2 == 2) {
    if (p.mode == 0) hipLaunchKernelGGL((conv_wino_kernel<0, 2>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_wino_kernel<1, 2>), grid, dim3(512), 0, st, p);
    MUNIT_CHECK_LAUNCH("conv_wino_s2_dgrad");
    return MUNIT_OK;
  }
  if (p.mode == 0) hipLaunchKernelGGL(conv_wino_kernel<0>, grid, dim3(512), 0, st, p);
  else if (p.mode == 1) hipLaunchKernelGGL(conv_wino_kernel<1>, grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL(conv_wino_kernel<2>, grid, dim3(512), 0, st, p);
  MUNIT_CHECK_LAUNCH("conv_wino");
  return MUNIT_OK;
}
