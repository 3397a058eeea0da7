/*
 * munit_hip.h -- C ABI of libmunit_hip.so: the gfx950 (MI355X / CDNA4) kernels behind the
 * MUNIT AdaINGen / AdaINGen_double + MsImageDis training step.
 *
 * The reference (cc-ai/MUNIT) is pure Python on PyTorch and has no FFI of its own
 * (SURVEY.md section 0.1); every entry point below therefore replaces a *stock torch op
 * call site* of the reference hot path, cited as scripts/<file>:<line>.  The Python side
 * (munit_amd/ops.py) binds these with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - All tensor pointers are DEVICE pointers.  Parameters, their gradients, statistics and loss scalars are float32;
 *     activations are float32 or -- in the bf16-storage mode, see MUNIT_DTYPE_* -- bfloat16, and then travel as void*.
 *     Activations are NHWC ([B][H][W][C], C contiguous) -- the memory image of a torch channels_last tensor.
 *     Convolution weights are [Cout][KH][KW][Cin] -- the memory image of a torch OIHW
 *     tensor in channels_last format, so state_dict shapes stay (O, I, KH, KW).
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*); the library
 *     never synchronises, allocates no device memory and keeps no state between calls
 *     (munit_stream_wait_stream caches one HIP event per host thread and device).
 *     The caller owns every buffer, including the workspace `ws` (size from the matching
 *     *_workspace_bytes query; must be 256-byte aligned).
 *   - Return value: 0 on success, negative on error; munit_last_error() returns a
 *     thread-local description of the last failure.
 */
#ifndef MUNIT_HIP_H
#define MUNIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* munit_stream_t; /* hipStream_t */

enum { MUNIT_OK = 0, MUNIT_ERR_ARG = -1, MUNIT_ERR_WORKSPACE = -2, MUNIT_ERR_LAUNCH = -3 };
enum { MUNIT_ACT_NONE = 0, MUNIT_ACT_RELU = 1, MUNIT_ACT_LRELU = 2, MUNIT_ACT_TANH = 3 };
enum { MUNIT_PAD_ZERO = 0, MUNIT_PAD_REFLECT = 1 };
/* MUNIT_COMPUTE_F32: exact fp32 MFMA (the reference's arithmetic).  MUNIT_COMPUTE_BF16: operands rounded to
 * bf16 (nearest-even) as they are staged into LDS, v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- the
 * bf16 mode of BASELINE.json config #3, a build extension with no reference counterpart.  Layers whose
 * channel count is not a multiple of 32 (the 3-channel image layers) stay fp32 in both modes.
 * MUNIT_COMPUTE_F32X3: fp32 accuracy on the bf16 matrix pipe -- each fp32 operand is split exactly into three
 * bf16 values (a = a0 + a1 + a2) and the six products a_i*b_j with i + j <= 2 are accumulated in fp32; the
 * dropped terms are <= 2^-24 |a||b|, below the rounding of an fp32 FMA chain.  Opt-in. */
enum { MUNIT_COMPUTE_F32 = 0, MUNIT_COMPUTE_BF16 = 1, MUNIT_COMPUTE_F32X3 = 2 };
/* Element type of an ACTIVATION tensor in HBM (BASELINE.json config #3: bf16 storage).  Parameters, biases, AdaIN
 * parameters, norm statistics, accumulators, gradients of parameters and loss scalars are always fp32.  A bf16 tensor
 * needs a channel count that is a multiple of 64 in the convolutions (one 128-byte tile row) and of 4 elsewhere;
 * 3-channel images and 1x1-spatial vectors stay fp32.  Pointers of such tensors travel as void*. */
enum { MUNIT_DTYPE_F32 = 0, MUNIT_DTYPE_BF16 = 1 };

int munit_version(void);
const char* munit_last_error(void);

/* Data-parallel exchange for hosts without a communicator of their own (no reference counterpart:
 * scripts/train.py:171-225 is single-process).  A thin layer over RCCL, resolved with dlopen at the first call (no link-time
 * dependency; inside a PyTorch process the already-loaded librccl is used).  Rank 0 calls munit_comm_unique_id and hands the
 * 128 bytes to every rank by its own means; every rank then calls munit_comm_init (collective), on the device it will use.
 * munit_comm_allreduce sums `count` floats in place over all ranks, asynchronously on `stream` (the flat gradient buffer of
 * an update; the caller scales by 1/world).  The Python host of this repository uses torch.distributed instead (the launch
 * contract of bench.py); these entry points are exercised by tests/test_gpu_dp.py at world size 1.
 * munit_shutdown releases what the library keeps between calls (the RCCL handle); it is refused with MUNIT_ERR_ARG while a
 * communicator made by munit_comm_init is still alive.  The first call from several host threads is serialised. */
typedef void* munit_comm_t;
int munit_comm_unique_id(void* id_out, size_t bytes);
int munit_comm_init(munit_comm_t* comm, int rank, int world, const void* unique_id);
int munit_comm_allreduce(munit_comm_t comm, float* buf, size_t count, munit_stream_t stream);
int munit_comm_destroy(munit_comm_t comm);
int munit_shutdown(void);

/* Stream plumbing (no reference counterpart: torch's autograd engine orders everything on one stream).  `waiter` waits
 * for all work enqueued so far on `signaler`; both streams belong to the current device.  Used to fork backward-weight
 * onto a side stream without creating torch Event / Stream objects per layer. */
int munit_stream_wait_stream(munit_stream_t waiter, munit_stream_t signaler);

/* ------------------------------------------------------------------------------------
 * Convolution.  Replaces nn.ReflectionPad2d/ZeroPad2d + nn.Conv2d (+ bias + activation)
 * of Conv2dBlock.forward (scripts/networks.py:695-701, pads :642-649, conv :691-693),
 * the bare nn.Conv2d heads (networks.py:68, :472), nn.Linear of LinearBlock
 * (networks.py:712, as a 1x1 convolution on a [B][1][1][K] image) and, with
 * upsample = 1, the nn.Upsample(scale_factor=2) + 5x5 conv pair of the decoder
 * (networks.py:532-546) without materialising the upsampled tensor.
 *   x: [B][H][W][Cin]   w: [Cout][KH][KW][Cin]   bias: [Cout] or NULL
 *   y: [B][Ho][Wo][Cout],  Ho = ((H << upsample) + 2*pad - KH) / stride + 1
 * act is applied after the bias (MUNIT_ACT_*; slope = LeakyReLU negative slope).
 * Implicit-GEMM on v_mfma_f32_16x16x4_f32 (exact fp32), LDS-staged NHWC tiles.
 * ------------------------------------------------------------------------------------ */
typedef struct {
  int B, H, W, Cin;
  int Cout, KH, KW;
  int stride, pad, pad_mode; /* MUNIT_PAD_* */
  int upsample;              /* 0 or 1: nearest x2 of x before padding */
  int act;                   /* MUNIT_ACT_* fused after bias (fwd only) */
  float slope;
  int compute;               /* MUNIT_COMPUTE_*: arithmetic of the contraction */
  int in_dtype, out_dtype;   /* MUNIT_DTYPE_* of x (and dx) / of y (and dy).  A bf16 x runs the bf16-storage kernels:
                              * direct-to-LDS tiles of 64 channels, v_mfma_f32_16x16x32_bf16, bf16 weight image */
} munit_conv_desc;

int munit_conv2d_out_hw(const munit_conv_desc* d, int* Ho, int* Wo);

size_t munit_conv2d_fwd_workspace_bytes(const munit_conv_desc* d); /* 0 for most layers */
int munit_conv2d_fwd(const munit_conv_desc* d, const void* x, const float* w, const float* bias,
                     void* y, void* ws, size_t ws_bytes, munit_stream_t stream);

/* backward-data (autograd of the sites above): dx[B][H][W][Cin] from dy[B][Ho][Wo][Cout]
 * (dy is the gradient w.r.t. the PRE-activation output; use munit_act_bwd first when an
 * activation was fused).  Handles the adjoint of reflect padding (border fold-add) and of
 * the nearest upsample (2x2 sum).  If add != NULL, dx = result + add (same shape). */
size_t munit_conv2d_dgrad_workspace_bytes(const munit_conv_desc* d);
int munit_conv2d_dgrad(const munit_conv_desc* d, const void* dy, const float* w, const void* add,
                       void* dx, void* ws, size_t ws_bytes, munit_stream_t stream);

/* backward-weight: dw = beta*dw + sum_m dy[m][co] * im2col(x)[m][kh][kw][ci], layout of w;
 * db = beta*db + sum_m dy[m][co] when db != NULL.  Deterministic split-K (slabs in ws). */
size_t munit_conv2d_wgrad_workspace_bytes(const munit_conv_desc* d);
int munit_conv2d_wgrad(const munit_conv_desc* d, const void* x, const void* dy, float* dw,
                       float* db, float beta, void* ws, size_t ws_bytes, munit_stream_t stream);

/* Prepared weight images.  Two passes multiply by a re-laid-out image of the layer's weights: backward-data
 * (flipped / transposed, one slice per stride phase: the autograd transpose of nn.Conv2d, networks.py:691-693) and the
 * sub-pixel forward of the nearest-x2 + 5x5 decoder convs (networks.py:532-546; four merged 3x3 phase kernels).
 * The fp32 3x3 / stride 1 / pad 1 layers with wide channel counts (the residual blocks, networks.py:603-624) and those
 * phase kernels run as Winograd F(2x2, 3x3) on the fp32 matrix pipe: their images are U = G g G^T (16 frequencies per
 * channel pair, laid out for the kernel's direct-to-LDS loads; _WINOGRAD_DGRAD: of the filter rotated by 180 degrees with
 * the channel roles swapped; _SUBPIXEL_WINOGRAD: of the four merged phase filters; _WINOGRAD_S2: of the four 2x2 parity
 * filters of a 4x4 / stride 2 layer, F(3x3, 2x2)).  Same results as the direct form to
 * fp32 rounding (the algorithm cuDNN uses for the reference's own fp32 3x3 convolutions).
 * Weights change only at optimizer.step() (scripts/trainer.py:252-268), so the caller may keep these images: query
 * the size (0 = the pass uses w as it is), fill a munit_prep_item on the host with munit_conv2d_prep_item, build
 * the image with munit_conv2d_prepare_weights (one layer) or munit_conv2d_prepare_weights_batch (a table of items
 * in DEVICE memory, one launch for every layer of an optimizer) and hand it to the *_prepared entry points.  With
 * wp == NULL those behave exactly like munit_conv2d_fwd / _dgrad (image rebuilt into the workspace per call). */
enum { MUNIT_PASS_FWD = 0, MUNIT_PASS_DGRAD = 1, MUNIT_PASS_WGRAD = 2 };
enum { MUNIT_PREP_NONE = 0, MUNIT_PREP_DGRAD = 1, MUNIT_PREP_SUBPIXEL = 2, MUNIT_PREP_CAST = 3,
       MUNIT_PREP_WINOGRAD = 4, MUNIT_PREP_WINOGRAD_DGRAD = 5, MUNIT_PREP_SUBPIXEL_WINOGRAD = 6,
       MUNIT_PREP_WINOGRAD_S2 = 7, MUNIT_PREP_WINOGRAD_S2_DGRAD = 8,
       MUNIT_PREP_SUBPIXEL_WINOGRAD_DGRAD = 9 /* [the _DGRAD image][Winograd image of the four rotated phase filters] */ };
typedef struct {
  const float* w; /* [Cout][KH][KW][Cin] */
  float* wp;      /* image, munit_conv2d_prepared_weight_bytes() bytes (bf16 elements when bf16 != 0) */
  int Cout, KH, KW, Cin;
  int kind;       /* MUNIT_PREP_*; _CAST = the weights as they are, rounded to bf16 (forward of a bf16-input layer) */
  int ps;         /* stride phases per axis (MUNIT_PREP_DGRAD) */
  int bf16;       /* image in bf16: the pass runs the bf16-storage kernels */
} munit_prep_item;
size_t munit_conv2d_prepared_weight_bytes(const munit_conv_desc* d, int pass);
int munit_conv2d_prep_item(const munit_conv_desc* d, int pass, const float* w, float* wp, munit_prep_item* out);
int munit_conv2d_prepare_weights(const munit_prep_item* item, munit_stream_t stream);
int munit_conv2d_prepare_weights_batch(const munit_prep_item* items_dev, int n, munit_stream_t stream);
int munit_conv2d_fwd_prepared(const munit_conv_desc* d, const void* x, const float* w, const void* wp,
                              const float* bias, void* y, void* ws, size_t ws_bytes, munit_stream_t stream);
int munit_conv2d_dgrad_prepared(const munit_conv_desc* d, const void* dy, const float* w, const void* wp,
                                const void* add, void* dx, void* ws, size_t ws_bytes, munit_stream_t stream);

/* nn.Linear of LinearBlock (scripts/networks.py:712, 743-749) under its own name: y[B][N] = act(x[B][K] w[N][K]^T + bias),
 * i.e. the 1x1 convolution on a [B][1][1][K] image (same kernels).  bwd: dx (or NULL), dw = beta*dw + dy^T x and
 * db likewise (or NULL); dy is the gradient at the PRE-activation output (apply munit_act_bwd first). */
size_t munit_linear_workspace_bytes(int B, int K, int N);
int munit_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, int act,
                     float slope, void* ws, size_t ws_bytes, munit_stream_t stream);
int munit_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int K,
                     int N, float beta, void* ws, size_t ws_bytes, munit_stream_t stream);

/* FLOPs (2 x multiply-accumulates over valid GEMM rows) the kernels ISSUE for one call of the pass, next to the
 * algorithmic 2*B*Ho*Wo*Cout*KH*KW*Cin of the torch op it replaces (nn.Conv2d / its autograd, networks.py:691-693):
 * the sub-pixel form of the up-sampling convs and the box-sum backward-data execute fewer, strided backward-data over
 * the padded domain and the 4-channel re-layout of the 3-channel image layers slightly more.  Measurement only
 * (bench.py: roofline.step_executed_tflop). */
double munit_conv2d_executed_flops(const munit_conv_desc* d, int pass);
/* Name, as rocprofv3 shows it, of the kernel (or kernel group) that carries `pass` of this layer: the dispatch of the three
 * entry points stated as text.  Measurement only (bench.py picks the dominant kernel of the step by measured time and names
 * it with this). Static storage; never NULL. */
const char* munit_conv2d_kernel_name(const munit_conv_desc* d, int pass);

/* dx = dy * act'(y) for the fused activations (y = post-activation output). n elements. */
int munit_act_bwd(int act, float slope, const float* y, const float* dy, float* dx, size_t n,
                  munit_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Instance norm / AdaIN.  Replaces nn.InstanceNorm2d(affine=False)
 * (scripts/networks.py:657) and AdaptiveInstanceNorm2d.forward = F.batch_norm on the
 * (1, B*C, H, W) view (networks.py:823-845): per-(b,c) mean and BIASED variance over HW,
 *   y = act( (x - mean) * rsqrt(var + eps) * weight[b][c] + bias[b][c] ) + residual
 * adain == NULL -> weight 1 / bias 0.  Otherwise weight[b][c] = adain[b*ad_ld + w_off + c],
 * bias[b][c] = adain[b*ad_ld + b_off + c] (the slicing of assign_adain_params,
 * networks.py:230-239, done by address).  relu: the activation act() as MUNIT_ACT_* -- 0 none, 1 ReLU,
 * 2 LeakyReLU(0.2), 3 tanh (networks.py:668-681 pairs any norm with any activation).  residual may be NULL
 * (ResBlock's `out += residual`, networks.py:620-624).
 * stats: [B][C][2] (mean, rstd) written for the backward.  C % 4 == 0.
 * ------------------------------------------------------------------------------------ */
size_t munit_instnorm_workspace_bytes(int B, int HW, int C);
int munit_instnorm_fwd(const float* x, float* y, float* stats, int B, int HW, int C,
                       const float* adain, int ad_ld, int w_off, int b_off, const float* residual,
                       int relu, float eps, void* ws, size_t ws_bytes, munit_stream_t stream);
/* dx from dy (gradient w.r.t. y before the residual add; the residual's gradient is dy
 * itself).  d_adain (same addressing as adain) receives dweight/dbias when not NULL. */
int munit_instnorm_bwd(const float* x, const float* dy, const float* stats, float* dx, int B, int HW,
                       int C, const float* adain, float* d_adain, int ad_ld, int w_off, int b_off,
                       int relu, void* ws, size_t ws_bytes, munit_stream_t stream);

/* bf16-storage forms (x, y, dy, dx, residual are bf16 tensors; statistics, AdaIN parameters and their gradients fp32) */
int munit_instnorm_fwd_bf16(const void* x, void* y, float* stats, int B, int HW, int C,
                            const float* adain, int ad_ld, int w_off, int b_off, const void* residual,
                            int relu, float eps, void* ws, size_t ws_bytes, munit_stream_t stream);
int munit_instnorm_bwd_bf16(const void* x, const void* dy, const float* stats, void* dx, int B, int HW,
                            int C, const float* adain, float* d_adain, int ad_ld, int w_off, int b_off,
                            int relu, void* ws, size_t ws_bytes, munit_stream_t stream);

/* ------------------------------------------------------------------------------------
 * MUNIT's custom LayerNorm (scripts/networks.py:851-878): per-sample mean and UNBIASED
 * std over C*H*W, y = act( (x - mean) / (std + eps) * gamma[c] + beta[c] ), act = MUNIT_ACT_* in `relu`.
 * stats: [B][2] (mean, std).  C % 4 == 0.
 * ------------------------------------------------------------------------------------ */
size_t munit_layernorm_workspace_bytes(int B, int HW, int C);
int munit_layernorm_fwd(const float* x, float* y, float* stats, int B, int HW, int C,
                        const float* gamma, const float* beta, int relu, float eps, void* ws,
                        size_t ws_bytes, munit_stream_t stream);
/* dgamma/dbeta: = acc*old + new (acc 0 or 1). */
int munit_layernorm_bwd(const float* x, const float* dy, const float* stats, float* dx, int B, int HW,
                        int C, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                        float acc, int relu, float eps, void* ws, size_t ws_bytes,
                        munit_stream_t stream);

/* bf16-storage forms (x, y, dy, dx bf16; gamma, beta and their gradients fp32) */
int munit_layernorm_fwd_bf16(const void* x, void* y, float* stats, int B, int HW, int C,
                             const float* gamma, const float* beta, int relu, float eps, void* ws,
                             size_t ws_bytes, munit_stream_t stream);
int munit_layernorm_bwd_bf16(const void* x, const void* dy, const float* stats, void* dx, int B, int HW,
                             int C, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                             float acc, int relu, float eps, void* ws, size_t ws_bytes,
                             munit_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Pooling.  nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False)
 * (scripts/networks.py:32-34) and nn.AdaptiveAvgPool2d(1) (networks.py:471).
 * ------------------------------------------------------------------------------------ */
int munit_avgpool3s2_fwd(const float* x, float* y, int B, int H, int W, int C, munit_stream_t stream);
int munit_avgpool3s2_bwd(const float* dy, float* dx, int B, int H, int W, int C, munit_stream_t stream);
int munit_gap_fwd(const float* x, float* y, int B, int HW, int C, munit_stream_t stream);
int munit_gap_bwd(const float* dy, float* dx, int B, int HW, int C, munit_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Losses.  torch.mean(torch.abs(a - b)) (scripts/trainer.py:290), the masked form
 * torch.mean(torch.abs((a - b) * (1 - mask))) (trainer.py:305; mask is [B][H][W], one value
 * per pixel, broadcast over C) and LSGAN torch.mean((x - target)**2) (networks.py:91,109).
 * out: one device float (written, deterministic two-stage reduction, partials in ws).
 * Backward: gout is a DEVICE scalar (upstream gradient, already times the loss weight).
 * ------------------------------------------------------------------------------------ */
size_t munit_loss_workspace_bytes(size_t n);
int munit_l1_mean_fwd(const float* a, const float* b, const float* mask, size_t npix, int C, float* out,
                      void* ws, size_t ws_bytes, munit_stream_t stream);
int munit_l1_mean_bwd(const float* a, const float* b, const float* mask, size_t npix, int C,
                      const float* gout, float* da, float* db, munit_stream_t stream);
/* recon_criterion on bf16 tensors (the content codes of the bf16-storage mode, trainer.py:470-471): a, b, da, db bf16 */
int munit_l1_mean_fwd_bf16(const void* a, const void* b, const float* mask, size_t npix, int C, float* out,
                           void* ws, size_t ws_bytes, munit_stream_t stream);
int munit_l1_mean_bwd_bf16(const void* a, const void* b, const float* mask, size_t npix, int C,
                           const float* gout, void* da, void* db, munit_stream_t stream);
int munit_mse_const_fwd(const float* x, float target, size_t n, float* out, void* ws, size_t ws_bytes,
                        munit_stream_t stream);
int munit_mse_const_bwd(const float* x, float target, size_t n, const float* gout, float* dx,
                        munit_stream_t stream);
/* out = sum_i w[i] * *(terms[i]); n <= 32; terms are device scalars, w host floats. */
int munit_weighted_sum(const float* const* terms, const float* w, int n, float* out,
                       munit_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Adam (torch.optim.Adam as configured at scripts/trainer.py:109-120: L2-coupled
 * weight_decay, amsgrad off) over one flat fp32 buffer of n elements.  step >= 1.
 * Hyper-parameters are doubles, as in torch: beta2 = 0.999 and 1 - beta2 = 0.001 are each rounded
 * to fp32 separately (computing 1 - (float)beta2 in fp32 is off by 1.3e-5 relative).
 * ------------------------------------------------------------------------------------ */
int munit_adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1,
                    double beta2, double eps, double weight_decay, int step, munit_stream_t stream);

/* ExtraAdam (scripts/extraadam.py:14-168; selected by `optimizer: extra...`, scripts/trainer.py:41-45,
 * stepped by the *_opt_step methods, trainer.py:252-268: extrapolation on even iterations, step on odd).
 * Every call advances the moments and forms u = -lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps).
 * mode 0: p_saved = p, p += u (first extrapolation since the last step); mode 1: p += u;
 * mode 2: p = p_saved + u (the update step). */
int munit_extraadam_step(float* p, const float* g, float* m, float* v, float* p_saved, size_t n, double lr,
                         double beta1, double beta2, double eps, double weight_decay, int step, int mode,
                         munit_stream_t stream);

/* y[i] = alpha * x[i] (+ y[i] if accumulate); used for the 1/world gradient averaging. */
int munit_scale(const float* x, float* y, size_t n, float alpha, int accumulate, munit_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Input pipeline.  Replaces the per-image torchvision/PIL chain of the
 * reference's loaders -- RandomHorizontalFlip -> Resize(new_size) -> RandomCrop -> ToTensor ->
 * Normalize(0.5, 0.5) (scripts/utils.py:229-249, 717-738; MyDataset.transform, utils.py:296-345) --
 * with one batched device pass over decoded uint8 images of different sizes.  The resize is Pillow's
 * BILINEAR resampler (anti-aliased, 22-bit fixed point, uint8 between the passes) restated bit-exactly.
 *   pool : device bytes holding the decoded images back to back (RGB, HWC interleaved; masks 1 byte/pixel)
 *   descs: device array of B descriptors (the random draws are made by the host loader)
 *   out  : images [B][out_h][out_w][3] fp32 in [-1, 1]; masks [B][out_h][out_w] fp32
 * ksize_max >= munit_image_ksize(src, rs) of every image axis in the batch (filter taps per output).
 * ------------------------------------------------------------------------------------ */
typedef struct {
  long long src_off;  /* byte offset of the image in pool */
  int src_h, src_w;   /* decoded size */
  int rs_h, rs_w;     /* size after Resize (= src size when no resize); unused for masks */
  int crop_i, crop_j; /* top-left corner of the crop window in the resized image */
  int flip;           /* 1: flipped left-right before the resize */
  int reserved;
} munit_image_desc;

int munit_image_ksize(int src_size, int rs_size);
size_t munit_image_preprocess_workspace_bytes(int B, int out_h, int out_w, int ksize_max);
int munit_image_preprocess(const unsigned char* pool, const munit_image_desc* descs, int B, int out_h,
                           int out_w, int ksize_max, float* out, void* ws, size_t ws_bytes,
                           munit_stream_t stream);
/* Mask chain of MyDataset.transform (utils.py:318-330): flip, NEAREST resize of the whole mask to
 * (out_w, out_h), crop of that image at (crop_j, crop_i) with zero fill past its edge (what the reference
 * computes), ToTensor, and x255 for samples whose maximum is 1. */
size_t munit_mask_preprocess_workspace_bytes(int B, int out_h, int out_w);
int munit_mask_preprocess(const unsigned char* pool, const munit_image_desc* descs, int B, int out_h,
                          int out_w, float* out, void* ws, size_t ws_bytes, munit_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MUNIT_HIP_H */
