// Winograd F(2x2, 3x3) convolution on the fp32 matrix pipe (conv_wino.hip): shared declarations.
#pragma once
#include "common.h"

// Transformed weight image of one layer (MUNIT_PREP_WINOGRAD / _WINOGRAD_DGRAD), as the kernel's fragment loads want
// it:  U[c = K/8][nb = N/64][f = 16][512]  floats, where K is the contraction channel (forward: Cin;
// backward-data: Cout), N the produced channel, f = 4*fi + fj the frequency of U = G g G^T; see wino_item_index for the
// order inside a plane.  Backward-data multiplies by the filter rotated by 180 degrees with the channel roles swapped.
__host__ __device__ inline long long wino_image_elems(int K, int N) { return 16ll * K * N; }

// (G g G^T)[fi][fj], G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
__device__ inline float wino_u_of_g(const float (&g)[3][3], int f) {
  const int fi = f >> 2, fj = f & 3;
  float t[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float a = g[0][s], b = g[1][s], cc = g[2][s];
    t[s] = fi == 0 ? a : fi == 1 ? 0.5f * ((a + cc) + b) : fi == 2 ? 0.5f * ((a + cc) - b) : cc;
  }
  return fj == 0 ? t[0] : fj == 1 ? 0.5f * ((t[0] + t[2]) + t[1]) : fj == 2 ? 0.5f * ((t[0] + t[2]) - t[1]) : t[2];
}

// Work item j of an image = one (contraction channel k, produced channel no) pair, all 16 frequencies: element
// ((j >> 9) * 16 + f) * 512 + (j & 511) for f = 0..15 (N = produced channels).  Inside a (chunk, N block, f) plane of
// 512 floats the order is the MFMA B-fragment order of the kernel: [half 2][lane 64][4], lane = 16 * kq + n holding
// (k = 2 kq, 2 kq + 1) of channel 16 * (2 half) + n, then of channel 16 * (2 half + 1) + n.
__device__ inline void wino_item_index(long long j, int N, int& k, int& no) {
  const int NB = N >> 6;
  const int w = (int)(j & 511);
  const int e = w & 1, ntl = (w >> 1) & 1, l = (w >> 2) & 63, h = w >> 8;
  const long long rest = j >> 9;
  const int nb = (int)(rest % NB), c = (int)(rest / NB);
  k = c * 8 + 2 * (l >> 4) + e;
  no = nb * 64 + (2 * h + ntl) * 16 + (l & 15);
}
__device__ inline void wino_store_item(float* __restrict__ img, long long j, const float (&g)[3][3]) {
  float* o = img + (j >> 9) * (16 * 512) + (j & 511);
#pragma unroll
  for (int f = 0; f < 16; ++f) o[f * 512] = wino_u_of_g(g, f);
}

// item j of the image of the OHWI fp32 weights w[Cout][3][3][Cin]; items = K * N
__device__ inline void wino_weight_item(const float* __restrict__ w, float* __restrict__ img, int Cout, int Cin, bool dgrad, long long j) {
  int k, no;
  wino_item_index(j, dgrad ? Cin : Cout, k, no);
  float g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      g[r][s] = dgrad ? w[(((long long)k * 3 + (2 - r)) * 3 + (2 - s)) * Cin + no] : w[(((long long)no * 3 + r) * 3 + s) * Cin + k];
  wino_store_item(img, j, g);
}

// 4x4 / stride 2 layers (conv_wino.hip, S2): contraction index k = phase * Cin + ci with phase = 2 r + s the parity of the
// filter tap (kh, kw) = (2a + r, 2b + s); the 2x2 filter of a phase is g[a][b] = w[no][2a + r][2b + s][ci], and
// U = G g G^T with G = [1 0; 1/2 1/2; 1/2 -1/2; 0 1].  w is [Cout][4][4][Cin]; items = 4 * Cin * Cout.
__device__ inline void wino_s2_weight_item(const float* __restrict__ w, float* __restrict__ img, int Cout, int Cin, long long j) {
  int k, no;
  wino_item_index(j, Cout, k, no);
  const int phase = k / Cin, ci = k - phase * Cin, r = phase >> 1, sft = phase & 1;
  float g[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) g[a][b] = w[(((long long)no * 4 + 2 * a + r) * 4 + 2 * b + sft) * Cin + ci];
  float t[4][2];   // G g
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b]);
    t[3][b] = g[1][b];
  }
  float* o = img + (j >> 9) * (16 * 512) + (j & 511);
#pragma unroll
  for (int fi = 0; fi < 4; ++fi) {
    o[(fi * 4 + 0) * 512] = t[fi][0];
    o[(fi * 4 + 1) * 512] = 0.5f * (t[fi][0] + t[fi][1]);
    o[(fi * 4 + 2) * 512] = 0.5f * (t[fi][0] - t[fi][1]);
    o[(fi * 4 + 3) * 512] = t[fi][1];
  }
}

// backward-data of a 4x4 / stride 2 layer: image [phase = 2 ph + pw][Cout / 8][Cin / 64][f][512]; the phase's 2x2 filter in
// patch order is g[a][b] = w[k][2 (1 - a) + ph][2 (1 - b) + pw][no] (k = output channel of the layer = contraction index)
__device__ inline void wino_s2_dgrad_weight_item(const float* __restrict__ w, float* __restrict__ img, int Cout, int Cin, long long j) {
  const long long per = (long long)Cin * Cout;
  const int phase = (int)(j / per), ph = phase >> 1, pw = phase & 1;
  const long long jj = j - phase * per;
  int k, no;
  wino_item_index(jj, Cin, k, no);
  float g[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) g[a][b] = w[(((long long)k * 4 + 2 * (1 - a) + ph) * 4 + 2 * (1 - b) + pw) * Cin + no];
  float t[4][2];   // G g
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b]);
    t[3][b] = g[1][b];
  }
  float* o = img + phase * wino_image_elems(Cout, Cin) + (jj >> 9) * (16 * 512) + (jj & 511);
#pragma unroll
  for (int fi = 0; fi < 4; ++fi) {
    o[(fi * 4 + 0) * 512] = t[fi][0];
    o[(fi * 4 + 1) * 512] = 0.5f * (t[fi][0] + t[fi][1]);
    o[(fi * 4 + 2) * 512] = 0.5f * (t[fi][0] - t[fi][1]);
    o[(fi * 4 + 3) * 512] = t[fi][1];
  }
}

// backward-data of a sub-pixel up-sampling layer (interior source pixels): contraction index k = phase * Cout + co with
// phase = 2a + b the output parity; filter of the phase in patch order = the merged 3x3 filter rotated by 180 degrees,
// g[r][s] = wm_ab[co][2 - r][2 - s][ci]  (wm as in wino_subpixel_weight_item).  One image: [K = 4 Cout][N = Cin].
__device__ inline void wino_subpixel_dgrad_weight_item(const float* __restrict__ w, float* __restrict__ img, int Cout, int Cin, long long j) {
  int k, no;
  wino_item_index(j, Cin, k, no);
  const int ph = k / Cout, co = k - ph * Cout, a = ph >> 1, b = ph & 1;
  float g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int sft = 0; sft < 3; ++sft) {
      const int dh = 2 - r, dw = 2 - sft;
      const int h0 = a == 0 ? (dh == 0 ? 0 : dh == 1 ? 2 : 4) : (dh == 0 ? 0 : dh == 1 ? 1 : 3);
      const int hn = a == 0 ? (dh == 2 ? 1 : 2) : (dh == 0 ? 1 : 2);
      const int w0 = b == 0 ? (dw == 0 ? 0 : dw == 1 ? 2 : 4) : (dw == 0 ? 0 : dw == 1 ? 1 : 3);
      const int wn = b == 0 ? (dw == 2 ? 1 : 2) : (dw == 0 ? 1 : 2);
      float sum = 0.f;
      for (int kh = h0; kh < h0 + hn; ++kh)
        for (int kw = w0; kw < w0 + wn; ++kw) sum += w[(((long long)co * 5 + kh) * 5 + kw) * Cin + no];
      g[r][sft] = sum;
    }
  wino_store_item(img, j, g);
}

// Sub-pixel form of nearest-x2-upsample + 5x5 conv (conv_igemm.hip, prep_subpixel_elem): phase (a, b) of the output is a
// 3x3 conv over the source with the 5 filter rows merged as  a=0: {0,1} {2,3} {4}   a=1: {0} {1,2} {3,4}  (columns
// likewise).  Image = [phase 4][the U image of that merged 3x3 filter]; w is [Cout][5][5][Cin]; items = 4 * Cin * Cout.
__device__ inline void wino_subpixel_weight_item(const float* __restrict__ w, float* __restrict__ img, int Cout, int Cin, long long j) {
  const long long per = (long long)Cin * Cout;
  const int ph = (int)(j / per), a = ph >> 1, b = ph & 1;
  int k, no;
  wino_item_index(j - ph * per, Cout, k, no);
  float g[3][3];
#pragma unroll
  for (int dh = 0; dh < 3; ++dh)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int h0 = a == 0 ? (dh == 0 ? 0 : dh == 1 ? 2 : 4) : (dh == 0 ? 0 : dh == 1 ? 1 : 3);
      const int hn = a == 0 ? (dh == 2 ? 1 : 2) : (dh == 0 ? 1 : 2);
      const int w0 = b == 0 ? (dw == 0 ? 0 : dw == 1 ? 2 : 4) : (dw == 0 ? 0 : dw == 1 ? 1 : 3);
      const int wn = b == 0 ? (dw == 2 ? 1 : 2) : (dw == 0 ? 1 : 2);
      float s = 0.f;
      for (int kh = h0; kh < h0 + hn; ++kh)
        for (int kw = w0; kw < w0 + wn; ++kw) s += w[(((long long)no * 5 + kh) * 5 + kw) * Cin + k];
      g[dh][dw] = s;
    }
  wino_store_item(img + ph * wino_image_elems(Cin, Cout), j - ph * per, g);
}

struct WinoParams {
  const float* x;      // NHWC input [B][H][W][K]
  const float* u;      // transformed weights (see above)
  const float* bias;   // [N] or null
  const float* add;    // null, or a tensor laid out like y that the epilogue adds (3x3 layers: the ResBlock skip gradient)
  float* y;            // output pixel (b, oh, ow) channel n at y[b*y_sb + oh*y_sh + ow*y_sw + n]
  long long y_sb, y_sh, y_sw;
  // gridDim.y launch phases (sub-pixel up-sampling conv: 4): phase ph = 2a + b reads u + ph * u_phase and writes to
  // y + a * y_prow + b * y_pcol
  long long u_phase, y_prow, y_pcol;
  int phases;
  unsigned x_bytes;    // extent of x for the buffer loads
  int B, H, W, K, N;   // K = contraction length (s2: 4 * xc), N = output channels; 3x3 layers: output extent = H x W
  int xc, cpp;         // channels per pixel of x; chunks of 8 channels per input phase (= K / 8 unless s2)
  int s2, Ho, Wo;      // 1: 4x4 / stride 2 / pad 1 layer forward (F(3x3, 2x2) over four input phases), output extent Ho x Wo;
                       // 2: its backward-data (x = dy of extent H x W, launch phases = parities, dx extent Ho x Wo)
  int mode;            // 0 reflect padding, 1 zero padding, 2 zero padding + border fold (backward-data of a reflect layer)
  int edge;            // mode 1, 3x3 layers: positions outside the image read the nearest edge pixel instead of 0
  int th, tw;          // 2x2 output tiles per image axis
  int bth, btw;        // 8x8-tile blocks per image axis
  int NB;              // N / 64
  int act;
  float slope;
};

// shapes the kernel takes (conv_wino.hip)
bool munit_wino_ok(int B, int H, int W, int K, int N);
int munit_wino_launch(const WinoParams& p, hipStream_t st);

// ---- backward-weight: dg = G^T [ sum_tiles (A dY A^T) . (B^T d B) ] G  (conv_wino.hip) ----
struct WinoWgradParams {
  const float* x;      // [B][H][W][Cin]
  const float* dy;     // output gradient; pixel (b, oh, ow) of the tiled region at dy_off + b*dy_sb + oh*dy_sh + ow*dy_sw
  float* slab;         // partial sums S[phase][split][f = 16][Cout][Cin]
  float* db_part;      // [phase][split][Cout] partial bias gradients, or null
  unsigned x_bytes, dy_bytes;
  long long dy_sb, dy_sh, dy_sw, dy_off;   // elements
  long long dy_prow, dy_pcol;              // launch phase ph = 2a + b (gridDim.y) adds a*dy_prow + b*dy_pcol to dy_off
  int B, H, W, Cin, Cout;
  int reflect;         // padding of x when the patch leaves the image (xo = -1): 1 reflect, 2 replicated edge, 0 zeros
  int ring_mask;       // launch phases = output parities (a, b) of an up-sampling layer: dy row 0 / column 0 of parity 0 and the
                       // last row / column of parity 1 (the outermost ring of the 2H x 2W output) read as 0
  int xo;              // patch origin: output pixel (oh, ow) reads x rows oh + xo .. oh + xo + 2 (-1: pad 1; 0: VALID)
  int th, tw, tiles;   // 2x2 output tiles per axis of the tiled region, in total
  int cps;             // chunks (of 8 tiles) per split
  int CB, NB, ksplit;  // Cin / 64, Cout / 64, splits of the tile range
  int phases;
  int s2, Ho, Wo;      // 4x4 / stride 2 layer: dy extent Ho x Wo, 3x3 tiles, launch phase = filter-tap parity (x phase)
};
bool munit_wino_wgrad_ok(int B, int H, int W, int Cin, int Cout);
// splits of the tile range, and the workspace (slabs + bias partials) for `phases` launch phases over `tiles` tiles
int munit_wino_wgrad_splits(long long tiles, int Cin, int Cout, int phases);
size_t munit_wino_wgrad_workspace(long long tiles, int Cin, int Cout, int phases);
// p: tensors, strides and geometry filled in (x, dy, *_bytes, dy_*, B..Cout, reflect, xo, th, tw, phases); writes
// dw + ph * dw_phase = beta * (.) + gradient of phase ph ([Cout][3][3][Cin]) and db = beta_b * db + sum over phases
int munit_wino_wgrad_launch(WinoWgradParams p, float* dw, long long dw_phase, float* db, float beta, float beta_b, void* ws,
                            hipStream_t st);
