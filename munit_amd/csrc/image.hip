// Input pipeline on the device: the per-image transform chain of the reference's loaders
//   RandomHorizontalFlip -> Resize(new_size) -> RandomCrop(h, w) -> ToTensor -> Normalize(0.5, 0.5)
// (scripts/utils.py:192-250, 680-740; MyDataset.transform utils.py:296-345) applied to a batch of
// decoded uint8 images of different sizes in one pass, writing the normalised NHWC float batch the
// convolutions read.  Only the crop window is ever computed.
//
// Resize arithmetic is Pillow's (the reference calls PIL through torchvision; requirements.txt pins
// Pillow==6.2.0, same resampler as today's): ImagingResample with the BILINEAR filter -- a separable
// anti-aliased triangle filter, coefficients computed in double and rounded to 22-bit fixed point,
// horizontal pass then vertical pass with a uint8 rounding between them -- restated here so that the
// result is bit-identical to Image.resize.  Masks use the NEAREST path (ImagingScaleAffine: source
// index tables from a running double accumulator).
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ inline double bilinear_filter(double x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return 1.0 - x;
  return 0.0;
}

__device__ inline int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One thread per (sample, axis, output index of the crop window): window start, length and the
// fixed-point coefficients of Pillow's precompute_coeffs + normalize_coeffs_8bpc.
//   tab layout per sample: [rows: out_h x (2 + ksize)] [cols: out_w x (2 + ksize)] ints
__global__ void resample_tables_kernel(const munit_image_desc* __restrict__ descs, int B, int out_h, int out_w,
                                       int ksize_max, int* __restrict__ tab) {
  const int per = (out_h + out_w) * (2 + ksize_max);
  const int total = B * (out_h + out_w);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / (out_h + out_w);
    const int r = i - b * (out_h + out_w);
    const munit_image_desc d = descs[b];
    const bool is_row = r < out_h;
    const int in_size = is_row ? d.src_h : d.src_w;
    const int rs_size = is_row ? d.rs_h : d.rs_w;
    const int xx = is_row ? d.crop_i + r : d.crop_j + (r - out_h);   // index in the resized image
    int* t = tab + (long long)b * per + (long long)r * (2 + ksize_max);

    const double scale = (double)in_size / (double)rs_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > ksize_max) xmax = ksize_max;   // cannot happen when the host sized ksize_max from the same formula
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bilinear_filter((x + xmin - center + 0.5) * ss);
    t[0] = xmin;
    t[1] = xmax;
    for (int x = 0; x < ksize_max; ++x) {
      int c = 0;
      if (x < xmax) {
        double w = bilinear_filter((x + xmin - center + 0.5) * ss);
        if (ww != 0.0) w /= ww;
        c = w < 0.0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
      }
      t[2 + x] = c;
    }
  }
}

// One thread per output pixel: horizontal pass over the rows of its vertical window (each rounded to
// uint8 as Pillow's intermediate image is), vertical pass, then ToTensor (/255) and Normalize
// ((t - 0.5) / 0.5) in fp32 -- the same operation order as torchvision, so the floats match bit for bit.
__global__ void image_resample_kernel(const unsigned char* __restrict__ pool,
                                      const munit_image_desc* __restrict__ descs, int B, int out_h, int out_w,
                                      int ksize_max, const int* __restrict__ tab, float* __restrict__ out) {
  const int per = (out_h + out_w) * (2 + ksize_max);
  const long long total = (long long)B * out_h * out_w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((long long)out_h * out_w));
    const int rem = (int)(i - (long long)b * out_h * out_w);
    const int y = rem / out_w, x = rem - y * out_w;
    const munit_image_desc d = descs[b];
    const unsigned char* src = pool + d.src_off;
    const int* ty = tab + (long long)b * per + (long long)y * (2 + ksize_max);
    const int* tx = tab + (long long)b * per + (long long)(out_h + x) * (2 + ksize_max);
    const int ymin = ty[0], ymax = ty[1], xmin = tx[0], xmax = tx[1];
    int v0 = 1 << (PRECISION_BITS - 1), v1 = v0, v2 = v0;
    for (int yy = 0; yy < ymax; ++yy) {
      const unsigned char* row = src + (long long)(ymin + yy) * d.src_w * 3;
      int h0 = 1 << (PRECISION_BITS - 1), h1 = h0, h2 = h0;
      for (int k = 0; k < xmax; ++k) {
        int sx = xmin + k;
        if (d.flip) sx = d.src_w - 1 - sx;
        const int c = tx[2 + k];
        h0 += row[sx * 3 + 0] * c;
        h1 += row[sx * 3 + 1] * c;
        h2 += row[sx * 3 + 2] * c;
      }
      const int cy = ty[2 + yy];
      v0 += clip8(h0) * cy;
      v1 += clip8(h1) * cy;
      v2 += clip8(h2) * cy;
    }
    float* o = out + i * 3;
    o[0] = (__fdiv_rn((float)clip8(v0), 255.f) - 0.5f) / 0.5f;
    o[1] = (__fdiv_rn((float)clip8(v1), 255.f) - 0.5f) / 0.5f;
    o[2] = (__fdiv_rn((float)clip8(v2), 255.f) - 0.5f) / 0.5f;
  }
}

// NEAREST index tables (ImagingScaleAffine): one thread per (sample, axis) walks xo += a0 in double.
//   tab layout per sample: [out_h row indices][out_w column indices] (-1 = no source pixel)
__global__ void nearest_tables_kernel(const munit_image_desc* __restrict__ descs, int B, int out_h, int out_w,
                                      int* __restrict__ tab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * B) return;
  const int b = i >> 1;
  const bool is_row = (i & 1) == 0;
  const munit_image_desc d = descs[b];
  const int in_size = is_row ? d.src_h : d.src_w;
  const int n = is_row ? out_h : out_w;
  int* t = tab + (long long)b * (out_h + out_w) + (is_row ? 0 : out_h);
  const double a = (double)in_size / (double)n;
  double xo = 0.0 + a * 0.5;
  for (int x = 0; x < n; ++x) {
    const int xin = xo < 0.0 ? -1 : (int)xo;
    t[x] = (xin >= 0 && xin < in_size) ? xin : -1;
    xo += a;
  }
}

// Mask chain of MyDataset.transform (utils.py:318-330): flip, NEAREST resize of the whole mask to the
// crop size (width, height), then crop((j, i, j+w, i+h)) of that image -- positions past its edge read 0
// (PIL crop semantics; the reference crops the already crop-sized mask at the image's offsets) -- and
// the per-sample maximum for the "max == 1 -> x255" rule.
__global__ void mask_gather_kernel(const unsigned char* __restrict__ pool, const munit_image_desc* __restrict__ descs,
                                   int B, int out_h, int out_w, const int* __restrict__ tab,
                                   float* __restrict__ out, int* __restrict__ vmax) {
  const long long total = (long long)B * out_h * out_w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((long long)out_h * out_w));
    const int rem = (int)(i - (long long)b * out_h * out_w);
    const int y = rem / out_w, x = rem - y * out_w;
    const munit_image_desc d = descs[b];
    const int my = y + d.crop_i, mx = x + d.crop_j;
    int v = 0;
    if (my < out_h && mx < out_w) {
      const int* t = tab + (long long)b * (out_h + out_w);
      const int sy = t[my];
      int sx = t[out_h + mx];
      if (sy >= 0 && sx >= 0) {
        if (d.flip) sx = d.src_w - 1 - sx;
        v = pool[d.src_off + (long long)sy * d.src_w + sx];
      }
    }
    out[i] = (float)v;
    if (v > 0) atomicMax(&vmax[b], v);
  }
}

__global__ void mask_scale_kernel(float* __restrict__ out, const int* __restrict__ vmax, int B, int hw) {
  const long long total = (long long)B * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / hw);
    float t = __fdiv_rn(out[i], 255.f);       // ToTensor
    if (vmax[b] == 1) t = t * 255.f;          // utils.py:326-329
    out[i] = t;
  }
}

int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 8192); }

}  // namespace

extern "C" int munit_image_ksize(int src_size, int rs_size) {
  if (src_size <= 0 || rs_size <= 0) return 0;
  double scale = (double)src_size / (double)rs_size;
  if (scale < 1.0) scale = 1.0;
  const double support = 1.0 * scale;
  return (int)ceil(support) * 2 + 1;
}

extern "C" size_t munit_image_preprocess_workspace_bytes(int B, int out_h, int out_w, int ksize_max) {
  return align_up((size_t)B * (out_h + out_w) * (2 + ksize_max) * sizeof(int), 256);
}

extern "C" int munit_image_preprocess(const unsigned char* pool, const munit_image_desc* descs, int B, int out_h,
                                      int out_w, int ksize_max, float* out, void* ws, size_t ws_bytes,
                                      munit_stream_t stream) {
  MUNIT_CHECK_ARG(pool && descs && out && ws, "image_preprocess: null pointer");
  MUNIT_CHECK_ARG(B > 0 && out_h > 0 && out_w > 0 && ksize_max >= 3, "image_preprocess: bad shape");
  if (ws_bytes < munit_image_preprocess_workspace_bytes(B, out_h, out_w, ksize_max)) {
    munit_set_error("image_preprocess: workspace %zu < %zu", ws_bytes,
                    munit_image_preprocess_workspace_bytes(B, out_h, out_w, ksize_max));
    return MUNIT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int* tab = reinterpret_cast<int*>(ws);
  hipLaunchKernelGGL(resample_tables_kernel, dim3(grid_for((long long)B * (out_h + out_w))), dim3(256), 0, st, descs, B,
                     out_h, out_w, ksize_max, tab);
  MUNIT_CHECK_LAUNCH("resample_tables");
  hipLaunchKernelGGL(image_resample_kernel, dim3(grid_for((long long)B * out_h * out_w)), dim3(256), 0, st, pool, descs,
                     B, out_h, out_w, ksize_max, tab, out);
  MUNIT_CHECK_LAUNCH("image_resample");
  return MUNIT_OK;
}

extern "C" size_t munit_mask_preprocess_workspace_bytes(int B, int out_h, int out_w) {
  return align_up((size_t)B * (out_h + out_w + 1) * sizeof(int), 256);
}

extern "C" int munit_mask_preprocess(const unsigned char* pool, const munit_image_desc* descs, int B, int out_h,
                                     int out_w, float* out, void* ws, size_t ws_bytes, munit_stream_t stream) {
  MUNIT_CHECK_ARG(pool && descs && out && ws, "mask_preprocess: null pointer");
  MUNIT_CHECK_ARG(B > 0 && out_h > 0 && out_w > 0, "mask_preprocess: bad shape");
  if (ws_bytes < munit_mask_preprocess_workspace_bytes(B, out_h, out_w)) {
    munit_set_error("mask_preprocess: workspace %zu < %zu", ws_bytes, munit_mask_preprocess_workspace_bytes(B, out_h, out_w));
    return MUNIT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int* tab = reinterpret_cast<int*>(ws);
  int* vmax = tab + (size_t)B * (out_h + out_w);
  if (hipMemsetAsync(vmax, 0, (size_t)B * sizeof(int), st) != hipSuccess) {
    munit_set_error("mask_preprocess: memset failed");
    return MUNIT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(nearest_tables_kernel, dim3(cdiv(2 * B, 64)), dim3(64), 0, st, descs, B, out_h, out_w, tab);
  MUNIT_CHECK_LAUNCH("nearest_tables");
  hipLaunchKernelGGL(mask_gather_kernel, dim3(grid_for((long long)B * out_h * out_w)), dim3(256), 0, st, pool, descs, B,
                     out_h, out_w, tab, out, vmax);
  MUNIT_CHECK_LAUNCH("mask_gather");
  hipLaunchKernelGGL(mask_scale_kernel, dim3(grid_for((long long)B * out_h * out_w)), dim3(256), 0, st, out, vmax, B,
                     out_h * out_w);
  MUNIT_CHECK_LAUNCH("mask_scale");
  return MUNIT_OK;
}
