// Winograd F(2x2, 3x3) convolution on the fp32 matrix pipe (conv_wino.hip): shared declarations.
#pragma once
#include "common.h"

// Transformed weight image of one layer (MUNIT_PREP_WINOGRAD / _WINOGRAD_DGRAD), as the kernel's direct-to-LDS loads
// want it:  U[c = K/8][nb = N/64][f = 16][n = 64][8]  floats, where K is the contraction channel (forward: Cin;
// backward-data: Cout), N the produced channel, f = 4*fi + fj the frequency of U = G g G^T, and the 8 channels of a
// chunk sit in the row as 4 pairs with pair q at slot q ^ (2 * ((n >> 3) & 1)) ^ ((n >> 4) & 3) -- the bank swizzle of the kernel's
// ds_read_b64 fragments.  Backward-data multiplies by the filter rotated by 180 degrees with the channel roles swapped.
__host__ __device__ inline long long wino_image_elems(int K, int N) { return 16ll * K * N; }

// element i of the image from the OHWI fp32 weights w[Cout][3][3][Cin]
__device__ inline float wino_weight_elem(const float* __restrict__ w, int Cout, int Cin, bool dgrad, long long i) {
  const int N = dgrad ? Cin : Cout;
  const int NB = N >> 6;
  const int kp = (int)(i & 7), n = (int)((i >> 3) & 63), f = (int)((i >> 9) & 15);
  const long long rest = i >> 13;
  const int nb = (int)(rest % NB), c = (int)(rest / NB);
  const int q = (kp >> 1) ^ (((n >> 3) & 1) << 1) ^ ((n >> 4) & 3);
  const int k = c * 8 + q * 2 + (kp & 1), no = nb * 64 + n;
  float g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      g[r][s] = dgrad ? w[(((long long)k * 3 + (2 - r)) * 3 + (2 - s)) * Cin + no] : w[(((long long)no * 3 + r) * 3 + s) * Cin + k];
  // row fi of G g (G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]), then column fj of (.) G^T
  const int fi = f >> 2, fj = f & 3;
  float t[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float a = g[0][s], b = g[1][s], cc = g[2][s];
    t[s] = fi == 0 ? a : fi == 1 ? 0.5f * ((a + cc) + b) : fi == 2 ? 0.5f * ((a + cc) - b) : cc;
  }
  return fj == 0 ? t[0] : fj == 1 ? 0.5f * ((t[0] + t[2]) + t[1]) : fj == 2 ? 0.5f * ((t[0] + t[2]) - t[1]) : t[2];
}

struct WinoParams {
  const float* x;      // NHWC input [B][H][W][K]
  const float* u;      // transformed weights (see above)
  const float* bias;   // [N] or null
  float* y;            // output pixel (b, oh, ow) channel n at y[b*y_sb + oh*y_sh + ow*y_sw + n]
  long long y_sb, y_sh, y_sw;
  unsigned x_bytes;    // extent of x for the buffer loads
  int B, H, W, K, N;   // output extent = input extent (3x3, stride 1, pad 1)
  int mode;            // 0 reflect padding, 1 zero padding, 2 zero padding + border fold (backward-data of a reflect layer)
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
  const float* dy;     // [B][H][W][Cout]
  float* slab;         // partial sums S[split][f = 16][Cout][Cin]
  float* db_part;      // [split][Cout] partial bias gradients, or null
  unsigned x_bytes, dy_bytes;
  int B, H, W, Cin, Cout;
  int reflect;
  int th, tw, tiles;   // 2x2 output tiles per image axis, in total
  int cps;             // chunks (of 8 tiles) per split
  int CB, NB, ksplit;  // Cin / 64, Cout / 64, splits of the tile range
};
bool munit_wino_wgrad_ok(int B, int H, int W, int Cin, int Cout);
// splits of the tile range for this shape, and the workspace (slabs + bias partials) they need
int munit_wino_wgrad_splits(int B, int H, int W, int Cin, int Cout);
size_t munit_wino_wgrad_workspace(int B, int H, int W, int Cin, int Cout);
int munit_wino_wgrad(const float* x, const float* dy, float* dw, float* db, float beta, int B, int H, int W, int Cin, int Cout,
                     int reflect, void* ws, hipStream_t st);
