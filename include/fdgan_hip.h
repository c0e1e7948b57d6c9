/* fdgan_hip.h -- C ABI of libfdgan_hip.so: the MI355X (gfx950) hot path of FD-GAN.
 *
 * The reference (WeilanAnnn/FD-GAN) has no FFI/plugin boundary: its hot path is
 * stock torch.nn ops called from models/dehaze1113.py, models/dehaze22.py and
 * myutils/vgg16.py.  This header is therefore the
 * boundary a maintainer binds INSTEAD of those torch.nn calls; every entry point
 * names the reference call site(s) it replaces.  INTEGRATION.md shows the ctypes
 * binding.
 *
 * Conventions
 *  - Plain C: raw device pointers, sizes, a hipStream_t passed as void*.  No torch
 *    or C++ types.  The caller owns every buffer; the library never allocates or
 *    frees device memory and retains no pointer after a call returns -- except
 *    inside an FdPlan, which retains the pointers recorded into it until destroyed.
 *  - All work is enqueued on the caller's stream, asynchronously, with no implicit
 *    synchronisation.  Re-entrant; one process per GPU in data-parallel runs.
 *  - Return value: FD_OK (0) or a negative FD_E* code; the message is available from
 *    fdgan_last_error() (thread-local).  Nothing throws or aborts across the ABI.
 *  - Two 16-bit element formats (FdTensor.dtype), same layout, strides and sizes:
 *      FD_F16  (IEEE fp16)  everything the FORWARD pass stores or multiplies: activations and the forward
 *                           filter images.  With bf16 storage the generator's output sits 46.8 dB from the
 *                           reference's fp32 CPU path whatever the kernels do; with fp16 (11-bit significand)
 *                           64 dB -- the 0.02 dB / 1e-3 metric budget needs >= 54 dB ("Precision").
 *      FD_BF16              activation GRADIENTS and the flipped filter images the data-gradient kernels multiply
 *                           them with: unscaled loss gradients (1e-7 per pixel) need fp32's exponent range.
 *    Both run on the MFMA pipe at the same rate (v_mfma_f32_16x16x32_f16 / _bf16), fp32 accumulation.  Every entry
 *    point checks the dtype of each view it is given; "activation" below means FD_F16, "gradient" FD_BF16.
 *  - Activations are NHWC 16-bit "views": element (n,h,w,c) lives at
 *    ptr + n*stride[0] + h*stride[1] + w*stride[2] + c (stride[3] must be 1), so a
 *    view can be a channel slice of a wider dense-block / concat buffer
 *    (torch.cat, dehaze1113.py:275,773,783,786, is never materialised).
 *    stride[2] (the pixel pitch) must be a multiple of 8 elements and ptr 16-byte
 *    aligned.  Network input / output may be NCHW fp32 (dtype FD_F32).
 *
 * The data-parallel gradient exchange (fdgan_allreduce_*; the reference's only multi-GPU mechanism is
 * nn.DataParallel, demo.py:89).  The training path keeps parameters and gradients in two flat fp32 buffers
 * (fdgan_adam_step below takes exactly those), so the exchange is a handful of large sums over slices of one
 * buffer.  The Python host of this repository issues them through torch.distributed -- RCCL on its own stream,
 * overlapped with the backward walk (fdgan_hip/optim.py: FlatAdam.overlap): the framework already owns the
 * communicator and the rendezvous.  A host that is NOT PyTorch uses the four entry points at the end of this
 * header: a thin wrapper of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy, RCCL looked up
 * with dlopen at the first call (no link-time dependency).
 */
#ifndef FDGAN_HIP_H
#define FDGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDGAN_ABI_VERSION 16

enum FdStatus {
  FD_OK = 0,
  FD_EINVAL = -1,       /* bad argument (shape, stride, alignment, NULL) */
  FD_EUNSUPPORTED = -2, /* valid request the library has no kernel for   */
  FD_ELAUNCH = -3,      /* HIP launch / runtime error                    */
  FD_ESTATE = -4        /* plan API misuse                               */
};

enum FdDtype { FD_BF16 = 0, FD_F32 = 1, FD_F16 = 2 };

enum FdAct {
  FD_ACT_NONE = 0,
  FD_ACT_RELU = 1,    /* nn.ReLU            dehaze1113.py:239,261; vgg16.py:28-46      */
  FD_ACT_LEAKY02 = 2, /* nn.LeakyReLU(0.2)  dehaze1113.py:34,213,221                   */
  FD_ACT_TANH = 3,    /* nn.Tanh            dehaze1113.py:799                          */
  FD_ACT_SIGMOID = 4  /* nn.Sigmoid         dehaze1113.py:223                          */
};

typedef void* FdStream; /* hipStream_t */

typedef struct FdTensor {
  void* ptr;
  int64_t n, h, w, c;
  int64_t stride[4]; /* element strides of n, h, w, c */
  int32_t dtype;     /* FdDtype */
  int32_t _pad;
} FdTensor;

/* Input-side fusion of a convolution: what the reference runs as separate modules
 * *before* the conv.  a = act(x * scale[c] + shift[c]) with
 *   scale = gamma / sqrt(var + eps), shift = beta - mean * scale
 * i.e. nn.BatchNorm2d followed by ReLU/LeakyReLU (torchvision _DenseLayer
 * norm1/relu1, norm2/relu2; _Transition norm/relu; dehaze1113.py:238-239).  Train
 * mode passes the batch statistics produced by fdgan_bn_finalize; eval mode passes
 * running_mean / running_var.  mean == NULL means no affine (plain activation, the
 * `*dy` blocks' in-place ReLU, dehaze1113.py:269,272,367).
 * pool2 != 0 additionally averages each 2x2 window of `a` (F.avg_pool2d(.,2),
 * dehaze1113.py:763,780; _Transition.pool commuted in front of the bias-free 1x1
 * conv) -- only valid for 1x1 convolutions.
 * If running_mean != NULL the launch also performs BatchNorm's train-mode side
 * effects for this norm: running <- (1-momentum)*running +
 * momentum*batch (unbiased variance via `count`), num_batches_tracked += 1. */
typedef struct FdPrologue {
  const float* mean;
  const float* var;
  const float* gamma;
  const float* beta;
  float eps;
  int32_t act;   /* FdAct: NONE, RELU or LEAKY02 */
  int32_t pool2; /* 0 / 1 */
  float momentum;
  float* running_mean;
  float* running_var;
  int64_t* num_batches_tracked;
  int64_t count; /* N*H*W of the normalised tensor (for the unbiased running_var) */
} FdPrologue;

typedef struct FdConvDesc {
  int32_t ksize;        /* 1, 3 or 4 (square)                                         */
  int32_t stride;       /* 1 or 2                                                     */
  int32_t pad;          /* zero padding, applied to the *activated* input             */
  int32_t epilogue_act; /* FdAct applied to conv(+bias) before it is stored           */
  int32_t upsample2;    /* 1: nearest x2 of the result (F.upsample_nearest :370)      */
  int32_t cout;         /* the filter's output channels (0: y->c); y->c may exceed it
                           up to the next multiple of 16 -- extra channels store zeros */
  int32_t w_layout;     /* FdWeightLayout the filter was packed with                  */
} FdConvDesc;

/* Per-output-channel batch statistics of what the conv stores (post bias/act,
 * pre rounding): the producer half of train-mode BatchNorm.  `partial` receives
 * rows x cpad x {sum, sum of squares} fp32; rows / cpad come from
 * fdgan_conv2d_fwd_info.  Reduce with fdgan_bn_finalize. */
typedef struct FdStats {
  float* partial;
  int64_t capacity_floats;
  /* Optional in-kernel finalize: when `mean` is set and the kernel supports it (FdConvInfo.fused_finalize), the
   * LAST workgroup to finish reduces the partial rows itself (fp64) and writes mean / biased variance of the
   * `cout` stored channels to mean[0..cout) / var[0..cout): no fdgan_bn_finalize launch.  `counter` is one
   * zero-initialised uint32 per stream of launches (the kernel resets it); `count` = N*H*W of the stored tensor. */
  float* mean;
  float* var;
  uint32_t* counter;
  int64_t count;
} FdStats;

typedef struct FdConvInfo {
  int64_t stats_rows;     /* rows the launch writes into FdStats.partial  */
  int64_t stats_cpad;     /* channels per row (Cout rounded up)           */
  int64_t grid_x, grid_y; /* for the record                               */
  int64_t lds_bytes;
  int64_t fused_finalize; /* 1: this kernel honours FdStats.mean / var / counter */
} FdConvInfo;

const char* fdgan_last_error(void);
int fdgan_version(void); /* == FDGAN_ABI_VERSION */
/* ABI v16: "<16 hex digits>[:flags]" -- the hash of the sources this library was compiled from (every .hip / .h of csrc/ and this
 * header; fdgan_hip/buildid.py), written at build time.  The Python binding refuses a library whose id differs from the hash
 * of the sources lying next to it: a stale build never runs. */
const char* fdgan_build_id(void);
/* ABI v16: CU budget of the CALLING THREAD's following launches (and dry runs): the persistent kernels -- one or two resident
 * workgroups per CU walking the work (conv1x1_ds, conv3x3_rs2 / _pw, conv1x1_xs, the streaming data gradients, the row-walking 3x3
 * weight gradient) -- size their grids for `ncu` CUs instead of the whole device, so that launches issued on a stream created
 * with a CU mask (hipExtStreamCreateWithCUMask) fill exactly that stream's share of the chip.  ncu <= 0: no budget.  Returns the
 * previous value.  Launches recorded into an FdPlan keep the grid they were recorded with. */
int fdgan_set_cu_budget(int ncu);
/* Name of the device the library sees (e.g. "gfx950"); NULL without a GPU. */
const char* fdgan_device_arch(void);

/* ---- weights ------------------------------------------------------------- */
/* Fragment orders of the packed 16-bit filter image (1 KiB per 16 cout x 32 k fragment):
 *   CHUNK32: k = 32-channel chunk-major, tap, then 8-channel groups (all kernels)
 *   X64    : 1x1 only; 64-channel k-steps whose lane groups own 16 consecutive channels,
 *            the order the x-stream kernel reads activations in (conv1x1_xs.hip).
 * fdgan_conv_weight_layout names the order the library wants for a given conv; pack
 * with it and pass it back in FdConvDesc.w_layout. */
enum FdWeightLayout { FD_WLAYOUT_CHUNK32 = 0, FD_WLAYOUT_X64 = 1 };
int fdgan_conv_weight_layout(int cout, int cin, int ksize, int stride);
/* Bytes of the packed 16-bit image of a (cout, cin, k, k) filter (upper bound over layouts). */
size_t fdgan_packed_weight_bytes(int cout, int cin, int ksize);
/* fp32 OIHW (nn.Conv2d.weight) or, with transposed != 0, IOHW
 * (nn.ConvTranspose2d.weight, dehaze1113.py:363 -- a 1x1 stride-1 transposed conv
 * is a 1x1 conv with the weight indexed (Cin,Cout)) -> packed image.
 * With flip != 0 the filter is additionally rotated 180 degrees and its in/out
 * channels swapped (the data-gradient filter).  dtype: FD_F16 for the image a forward
 * convolution multiplies activations with, FD_BF16 for the image a data-gradient
 * convolution multiplies gradients with. */
int fdgan_pack_conv_weight(const float* w, int cout, int cin, int ksize, int transposed, int flip,
                           int layout, int dtype, void* packed, size_t packed_bytes, FdStream stream);

/* Batched form for a whole network: `jobs` is a table in DEVICE memory (it is read by the kernel), one entry per packed
 * image -- e.g. every filter of a generator in both orientations -- and the launch packs all of them at once (one
 * launch per optimizer step instead of one per filter and orientation).  first_unit = running sum of
 * fdgan_pack_units(cout, cin, ksize, layout) over the preceding jobs; total_units = that sum over all jobs.  Same
 * semantics per job as fdgan_pack_conv_weight. */
typedef struct FdPackJob {
  const float* w;
  void* packed;
  int32_t cout, cin, ksize, transposed, flip, layout;
  int32_t dtype; /* FD_F16 (forward image) or FD_BF16 (gradient-side image) */
  int32_t _pad;
  int64_t first_unit;
} FdPackJob;
int64_t fdgan_pack_units(int cout, int cin, int ksize, int layout);
int fdgan_pack_conv_weights(const FdPackJob* jobs_device, int64_t njobs, int64_t total_units, FdStream stream);

/* ---- convolution ---------------------------------------------------------- */
/* Replaces nn.Conv2d / nn.ConvTranspose2d(1x1) forward and the BN/ReLU/pool/cat/
 * upsample/tanh/sigmoid modules fused around it:
 *   y = act_e( conv_k,s,p( pool?( act_p( bn?(x) ) ) ) + bias )   [nearest x2]
 * x: NHWC fp16 view with c = Cin.  y: NHWC fp16 view, or NCHW fp32, with c = Cout.
 * (x bf16: a plain stride-1 convolution over GRADIENTS with a bf16 filter image -- the unfused data gradient; y bf16.)
 * w_packed: from fdgan_pack_conv_weight(cout, cin, ksize).  bias: fp32[Cout] or NULL.
 * Call sites replaced: torchvision _DenseLayer.conv1/conv2, _Transition.conv,
 * dehaze1113.py:262,265,363 (dy blocks), :744-755 (refine convs), :196-222 (D),
 * vgg16.py:9-21. */
int fdgan_conv2d_fwd_info(const FdTensor* x, const FdTensor* y, int cout, const FdConvDesc* d,
                          const FdPrologue* pro, FdConvInfo* info);
int fdgan_conv2d_fwd(const FdTensor* x, const void* w_packed, const float* bias,
                     const FdPrologue* pro, const FdTensor* y, const FdStats* stats,
                     const FdConvDesc* d, FdStream stream);

/* ---- batch-norm statistics -------------------------------------------------- */
/* partial[rows][cpad][2] -> mean[c], var[c] (biased, as nn.BatchNorm2d normalises
 * with in train mode), c < channels; `count` = N*H*W. */
int fdgan_bn_finalize(const float* partial, int64_t rows, int64_t cpad, int64_t channels,
                      int64_t count, float* mean, float* var, FdStream stream);

/* ---- layout helpers --------------------------------------------------------- */
/* NCHW fp32 (n,c,h,w contiguous) -> NHWC fp16 / bf16 view y, rounded to y->dtype (y->c >= c; channels
 * c..y->c-1 are written as zeros). */
int fdgan_nchw_f32_to_nhwc(const float* x, int64_t n, int64_t c, int64_t h, int64_t w,
                           const FdTensor* y, FdStream stream);
/* NHWC fp16 / bf16 view -> NCHW fp32 contiguous. */
int fdgan_nhwc_to_nchw_f32(const FdTensor* x, float* y, FdStream stream);
/* Channel-slice copy between NHWC views of one 16-bit format and equal n,h,w,c (c multiple of 8). */
int fdgan_copy_nhwc(const FdTensor* src, const FdTensor* dst, FdStream stream);

/* ---- legacy DCPDN-era networks (/root/reference/models/dehaze22.py G :205-362, G2 :364-488,
 * Dense :531-660; models/dehaze1113.py Dense :431-570).  Everything else those networks need is fdgan_conv2d_fwd: the
 * ConvTranspose2d(4, 2, 1) of blockUNet (dehaze22.py:60) is four stride-1 3x3 convolutions, one per output parity,
 * writing through strided views.
 *
 * fdgan_pyramid_pool4: the multi-scale head, dehaze22.py:343-356 / :634-651.  For the four windows k0, k0/2, k0/4, k0/8
 * (k0 = 16 or 32): avg_pool2d(x, k) -> Conv2d(C, 1, 1)(weight[j], bias[j]) -> LeakyReLU(slope) -> upsample_nearest to
 * x's size; the four maps land in the 4-channel view y (typically a slice of the buffer the next conv reads).  H and W
 * must be multiples of k0.  weight: [4][C] fp32, bias: [4] fp32.
 *
 * fdgan_bn_dropout_nhwc: y = mask[n][c] * ((x - mean[c]) / sqrt(var[c] + eps) * gamma[c] + beta[c]) on NHWC fp16 views:
 * train-mode nn.BatchNorm2d followed by train-mode nn.Dropout2d (dehaze22.py:60-63; the caller draws the (N, C) mask of
 * 0 / 1/(1-p) values).  mean == NULL: no normalisation; gamma / beta == NULL: 1 / 0; mask == NULL: no dropout.  Padding
 * channels of the 8-channel groups are written as zero. */
/* fdgan_scatter_dehaze: the arithmetic between the sub-networks of `dehaze` (dehaze22.py:699-715), all tensors contiguous
 * N x 3 x H x W fp32:  A = upsample_nearest(LeakyReLU(slope)(avg_pool2d(atp, H)), (H, W))  (window_mean: scratch of
 * N * 3 * (W / H) floats),  J = (x - A) / (|tran| + eps) + A.  Outputs: atp_out = A, dehaze2 = J (two of the network's four
 * return values) and `cat`, an N x H x W x 8 NHWC fp16 view receiving [J (3) | x (3) | 0 0], the input of refine1. */
int fdgan_scatter_dehaze(const float* x, const float* tran, const float* atp, int64_t n, int64_t h, int64_t w, float slope, float eps,
                         float* window_mean, float* atp_out, float* dehaze2, const FdTensor* cat, FdStream stream);
/* fdgan_maxpool3s2_nhwc: y = MaxPool2d(kernel 3, stride 2, padding 1)(act(bn(x))) -- torchvision DenseNet-121's norm0 / relu0 /
 * pool0 as `Dense` uses them (dehaze22.py:540-543, :607).  `pro` as for a convolution (mean == NULL: no normalisation; act NONE or
 * RELU; the running-statistics side effect is NOT applied here).  partial != NULL: one row of per-channel (sum, sum of squares)
 * of the stored values per workgroup, rows_out rows of y->c channels, for fdgan_bn_finalize. */
int fdgan_maxpool3s2_nhwc(const FdTensor* x, const FdPrologue* pro, const FdTensor* y, float* partial, int64_t capacity_floats,
                          int64_t* rows_out, FdStream stream);
int fdgan_pyramid_pool4(const FdTensor* x, const float* weight, const float* bias, int k0, float slope, const FdTensor* y,
                        FdStream stream);
int fdgan_bn_dropout_nhwc(const FdTensor* x, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                          const float* mask, const FdTensor* y, FdStream stream);
/* Reverse mode of the four entry points above (csrc/legacy_bwd.hip; torch.autograd through dehaze22.py:343-356, :540-543, :60-63,
 * :699-715).  Gradients are NHWC bf16 views like every activation gradient.
 * fdgan_maxpool3s2_bwd: da (N x H x W x C, written) = gradient w.r.t. act(bn(x)): each position collects dy of the windows whose
 *   FIRST maximum (scan order, ATen's rule) it is; continue with fdgan_bn_act_bwd / fdgan_bn_bwd_finalize / fdgan_bn_bwd_apply.
 * fdgan_pyramid_pool4_bwd: dx += the head's input gradient; dw_part [tiles][4][C] / db_part [tiles][4] = per-tile partial sums of
 *   the four 1x1 filters' gradients (tiles = N * H/k0 * W/k0, returned in *tiles_out; sum them over the first axis).
 * fdgan_bn_dropout_bwd: x = the RAW values the forward normalised, dy = gradient of mask * bn(x); dx written, dgamma / dbeta [C]
 *   written (batch statistics: the full BatchNorm backward).  mean == NULL: dx = mask * dy.
 * fdgan_scatter_dehaze_bwd: gradients w.r.t. the transmission map and G2's output (both N x 3 x H x W fp32, written) from the
 *   gradients of `dehaze2` and `atp` (fp32 NCHW, either may be NULL) and of the refine input's J channels (g_cat channels 0-2,
 *   may be NULL); window_mean as fdgan_scatter_dehaze left it; scratch >= N * 3 * (W / H) * H floats. */
int fdgan_maxpool3s2_bwd(const FdTensor* x, const FdPrologue* pro, const FdTensor* dy, const FdTensor* da, FdStream stream);
int fdgan_pyramid_pool4_bwd(const FdTensor* x, const float* weight, const float* bias, int k0, float slope, const FdTensor* dy,
                            const FdTensor* dx, float* dw_part, float* db_part, int64_t* tiles_out, FdStream stream);
int fdgan_bn_dropout_bwd(const FdTensor* x, const float* mean, const float* var, const float* gamma, float eps, const float* mask,
                         const FdTensor* dy, const FdTensor* dx, float* dgamma, float* dbeta, FdStream stream);
int fdgan_scatter_dehaze_bwd(const float* x, const float* tran, const float* atp, const float* window_mean, int64_t n, int64_t h,
                             int64_t w, float slope, float eps, const float* g_dehaze2, const float* g_atp, const FdTensor* g_cat,
                             float* d_tran, float* d_atp, float* scratch, int64_t scratch_floats, FdStream stream);

/* Element-wise dropout of the dy blocks (dropRate > 0: F.dropout at /root/reference/models/dehaze1113.py:270-274, :367-368; FDGAN
 * itself constructs them with 0): dst *= mask in place, mask an NHWC fp16 tensor holding 0 or 1 / (1 - p) that the host drew.  dst:
 * the fp16 activation (forward) or the bf16 gradient of it (backward: the same multiply).  up2 != 0: dst is twice the mask's
 * size, pixel (y, x) takes mask (y / 2, x / 2) -- TransitionBlockdy drops before its nearest x2 upsample. */
int fdgan_mul_mask_nhwc(const FdTensor* mask, const FdTensor* dst, int up2, FdStream stream);

/* ---- plan: record once, replay many ----------------------------------------- */
/* Between fdgan_plan_begin and fdgan_plan_end every launching entry point above,
 * called from the same thread, is recorded into the plan instead of being
 * enqueued.  fdgan_plan_launch replays the recorded kernels in order on `stream`
 * (as a hipGraph once fdgan_plan_instantiate_graph succeeded).  The plan retains
 * the recorded device pointers: keep those buffers alive and in place. */
typedef struct FdPlan FdPlan;
FdPlan* fdgan_plan_create(void);
void fdgan_plan_destroy(FdPlan* p);
int fdgan_plan_begin(FdPlan* p);
int fdgan_plan_end(FdPlan* p);
int64_t fdgan_plan_num_launches(const FdPlan* p);
int fdgan_plan_launch(FdPlan* p, FdStream stream);
int fdgan_plan_instantiate_graph(FdPlan* p, FdStream stream);
/* Multi-stream recording (round 4: the REVERSE walk of a network is recorded too, and it runs its unfused weight gradients on a
 * second stream beside the chain of data gradients -- the reference's counterpart is torch.autograd's engine walking
 * /root/reference/models/dehaze1113.py:758-801 backwards).  While a plan records, fdgan_plan_set_slot names the stream slot
 * (0 .. 7; 0 = the launch stream) the following launches belong to, and fdgan_plan_record_wait records a dependency: everything
 * recorded afterwards for `waiter_slot` runs after everything recorded so far for `signaler_slot` (an event recorded on one
 * stream and waited for on the other at replay).  fdgan_plan_launch_multi replays such a plan on the caller's streams
 * (streams[slot]); fdgan_plan_launch / _instantiate_graph / _profile refuse plans that use more than slot 0. */
int fdgan_plan_set_slot(FdPlan* p, int slot);
int fdgan_plan_record_wait(FdPlan* p, int waiter_slot, int signaler_slot);
int fdgan_plan_launch_multi(FdPlan* p, const FdStream* streams, int nstreams);
/* Name of the k-th recorded kernel (for profiles/tests; "stream_wait" for a recorded dependency); NULL if out of range. */
const char* fdgan_plan_kernel_name(const FdPlan* p, int64_t k);
/* Measurement (bench.py): mark launches (strictly increasing indices; n == 0 clears).
 * While marks exist fdgan_plan_launch replays eagerly and brackets every marked launch
 * with a hipEvent pair recorded on the launch stream.  fdgan_plan_read_timing waits
 * for the recorded pairs, returns their summed elapsed milliseconds and count, and
 * resets.  fdgan_plan_profile replays once with an event after every launch,
 * synchronises the stream and fills ms_out[n == number of launches]. */
int fdgan_plan_time_launches(FdPlan* p, const int64_t* idx, int64_t n);
int fdgan_plan_read_timing(FdPlan* p, double* total_ms, int64_t* launches);
int fdgan_plan_profile(FdPlan* p, FdStream stream, float* ms_out, int64_t n);

/* ---- pooling ------------------------------------------------------------------- */
/* ---- backward (first version: D's training path; the generator's dense blocks follow) -------------
 * Autograd of the fused forward op  a = act(bn(x)); y = conv(a, W) + b  (nn.Conv2d / nn.BatchNorm2d /
 * nn.LeakyReLU as composed in dehaze1113.py:188-230).  Gradients of activations are NHWC bf16 views laid out
 * like the (fp16) activations themselves; parameter gradients are fp32 in the parameter's own layout.  In every
 * signature below x / fwd_x / ref (a forward activation) is FD_F16 and dy / da / dpre / dx / g (a gradient) FD_BF16.
 *
 *  data gradient    da = conv^T(dy, W): call fdgan_conv2d_fwd on dy with the filter packed by
 *                   fdgan_pack_conv_weight(w, cout' = cin, cin' = cout, k, 0, flip = 1, layout, FD_BF16, ...) and pad' = k-1-pad
 *                   (stride 1); fdgan_conv2d_bwd_data_direct covers any stride for gradients w.r.t.
 *                   network inputs (NCHW fp32, few channels).
 *  weight gradient  fdgan_conv2d_bwd_weight: dW[co][ci][ky][kx] = sum_px dy[px][co] * a[px*s + k - pad][ci],
 *                   `a` recomputed from the raw input x and the forward prologue (batch statistics, no side
 *                   effects); dbias = sum_px dy (or NULL).
 *  prologue         fdgan_bn_act_bwd:  dpre = da * act'(bn(x)) in place, plus rows x cpad x {sum dpre,
 *                   sum dpre*xhat} partials when the prologue has a norm -> fdgan_bn_bwd_finalize -> (dgamma,
 *                   dbeta); fdgan_bn_bwd_apply: dx = gamma*rstd * (dpre - dbeta/M - xhat*dgamma/M).
 *                   With a workspace the pixel axis is split over workgroups and the partials are summed in a
 *                   fixed order (deterministic); `accumulate` adds into dw / dbias (parameters used twice).
 *                   A pooled prologue (pool2) is honoured; for an upsampled output pass the 2x2-summed gradient.
 *  output act       fdgan_out_act_bwd: g (NHWC bf16) = dout * f'(out) for the NCHW fp32 tensors the networks
 *                   return (sigmoid map of D: s(1-s); tanh image of FDGAN: 1-t^2).
 *  plumbing         fdgan_grad_ew: 0 dst += src (several consumers of one tensor, torch.cat), 1 dst = src[y/2][x/2]/4
 *                   (prologue avg-pool), 2 dst = 2x2 sum of src (nearest-upsample epilogue), 3 dst = src * (ref > 0)
 *                   (ReLU epilogue, ref = the stored output), 4 dst = src * (ref > 0 ? 1 : 0.2) (LeakyReLU epilogue). */
int fdgan_conv2d_bwd_weight(const FdTensor* x, const FdPrologue* pro, const FdTensor* dy, const FdConvDesc* d,
                            float* dw, float* dbias, float* workspace, int64_t workspace_floats, int accumulate,
                            FdStream stream);
int fdgan_bn_act_bwd(const FdTensor* da, const FdTensor* x, const FdPrologue* pro, float* partial,
                     int64_t capacity_floats, int64_t* rows_out, int64_t* cpad_out, FdStream stream);
/* fdgan_bn_act_bwd for a POOLED prologue that also adds gamma * rstd * dpre into dx (the full-resolution gradient of x) in the same
 * pass: with the sums reduced (fdgan_bn_bwd_finalize) and turned into the per-channel remainder B * x + C (fdgan_bn_bwd_coef), which a
 * caller that defers such remainders (fdgan_affine_accumulate) already knows how to apply, the second pass over the full-resolution
 * input -- fdgan_bn_bwd_apply -- is not needed (torchvision _Transition under autograd, dehaze1113.py:716-728). */
int fdgan_bn_act_bwd_acc(const FdTensor* da, const FdTensor* x, const FdPrologue* pro, const FdTensor* dx, float* partial,
                         int64_t capacity_floats, int64_t* rows_out, int64_t* cpad_out, FdStream stream);
/* The same with dx_store != 0: dx = gamma * rstd * dpre instead of +=, for the FIRST writer of a gradient buffer in a backward walk
 * when it covers the whole buffer (a dense block's transition, the last forward reader of the block's concat buffer: torchvision
 * _Transition, dehaze1113.py:716-728): the buffer then needs no zeroing at the start of the walk and is not read here. */
int fdgan_bn_act_bwd_dx(const FdTensor* da, const FdTensor* x, const FdPrologue* pro, const FdTensor* dx, int dx_store, float* partial,
                        int64_t capacity_floats, int64_t* rows_out, int64_t* cpad_out, FdStream stream);
int fdgan_bn_bwd_finalize(const float* partial, int64_t rows, int64_t cpad, int64_t channels, float* dgamma,
                          float* dbeta, int accumulate, FdStream stream);
/* Same, and the sums are ALSO added into sink_dgamma / sink_dbeta when non-NULL: the BatchNorm parameters' own
 * gradient buffers (an optimizer's flat gradient), so a training step needs no per-parameter add kernels. */
int fdgan_bn_bwd_finalize_sink(const float* partial, int64_t rows, int64_t cpad, int64_t channels, float* dgamma,
                               float* dbeta, int accumulate, float* sink_dgamma, float* sink_dbeta, FdStream stream);
/* (dgamma, dbeta) from RAW moments: partial rows of (sum dpre, sum dpre * x) as fdgan_conv2d_bwd_data writes them;
 * dgamma = (S2 - mean * S1) / sqrt(var + eps), dbeta = S1.  sink_*: as above.  scratch (optional, >= 64 * cpad floats):
 * lets more than 256 rows be reduced in two levels (32 row slices in parallel, then one row per slice). */
int fdgan_bn_bwd_finalize_raw(const float* partial, int64_t rows, int64_t cpad, int64_t channels, const float* mean,
                              const float* var, float eps, float* dgamma, float* dbeta, float* sink_dgamma,
                              float* sink_dbeta, float* scratch, int64_t scratch_floats, FdStream stream);
/* fdgan_bn_bwd_finalize_raw followed by fdgan_bn_bwd_coef (below) in ONE launch: the sums go to the parameters' gradient
 * sinks (optional) and into the coefficient pair (bsum, csum) of the deferred affine; pro: the forward prologue (mean, var,
 * gamma, eps), count = N*H*W of the normalised tensor.  coef_store != 0 (ABI v12): bsum / csum are WRITTEN instead of added to --
 * for a tensor with a single normalising consumer (a dense layer's bottleneck) the pair then never needs zeroing. */
int fdgan_bn_bwd_finalize_coef(const float* partial, int64_t rows, int64_t cpad, int64_t channels, const FdPrologue* pro,
                               int64_t count, float* sink_dgamma, float* sink_dbeta, float* bsum, float* csum, float* scratch,
                               int64_t scratch_floats, int coef_store, FdStream stream);
/* Data gradient of a stride-1 conv (reference: autograd of nn.Conv2d as composed in models/dehaze1113.py:188-230,
 * :703-801) fused with the first backward pass of the conv's input-side prologue: runs the FORWARD kernel on dy with
 * the flipped filter (fdgan_pack_conv_weight(..., flip = 1), d->pad = k - 1 - pad) and stores
 *     dpre = conv^T(dy, W) * act'(bn(fwd_x))
 * where fwd_x is the forward conv's raw input and fwd_pro its prologue (BatchNorm batch statistics + activation, no
 * pooling; NULL: identity).  With a norm, `partial` receives rows x cpad x {sum dpre, sum dpre * fwd_x} per channel
 * (rows / cpad returned) for fdgan_bn_bwd_finalize_raw; fdgan_bn_bwd_apply completes dx.  Replaces
 * fdgan_conv2d_fwd + fdgan_bn_act_bwd: one pass less over the gradient tensor.
 * accumulate = 1: `dpre` is the GRADIENT BUFFER of fwd_x instead and receives dx += gamma * rstd * dpre (dx += dpre
 * without a norm); dpre itself is never stored.  accumulate = 2: the same value is STORED (dx = ...): for an input with
 * this conv as its only consumer, whose gradient buffer then needs neither zeroing nor reading.  The rest of BatchNorm's backward, B * x + C per channel, is linear in x:
 * fdgan_bn_bwd_coef adds a layer's (B, C) into a coefficient pair shared by every layer that normalises those channels
 * (the layers of a dense block) and one fdgan_affine_accumulate pass, dx += Bsum * x + Csum, serves them all. */
int fdgan_conv2d_bwd_data(const FdTensor* dy, const void* w_packed_flipped, const FdTensor* fwd_x, const FdPrologue* fwd_pro,
                          const FdTensor* dpre, int accumulate, float* partial, int64_t capacity_floats, int64_t* rows_out,
                          int64_t* cpad_out, const FdConvDesc* d, FdStream stream);

/* The dense-layer bottleneck (1x1 conv, 128 filters, torchvision _DenseLayer.conv1 behind norm1 / relu1, as used at
 * /root/reference/models/dehaze1113.py:713-724) backward in ONE pass: fdgan_conv2d_bwd_data's result (same arguments, same
 * partial sums for BatchNorm) AND the weight gradient dw[128][C] (+= when dw_accumulate) of the same conv, whose operands --
 * the dy tile and act(bn(x)) -- are on chip anyway.  Saves the weight-gradient kernel's read of dy and x (a quarter of the
 * pair's HBM traffic).  wgrad_workspace: fp32 scratch for one [128][C] partial per pixel slot (<= 512 / ceil(C / 128)
 * slots), summed in a fixed order.  dy must have 128 channels, N*H*W a multiple of 64, views dense; FD_EUNSUPPORTED
 * otherwise (nothing launched: call the two separate entry points).
 * dy_affine_x != NULL: dy is not final yet -- the linear remainder of the BatchNorm backward of dy's own producer (norm2 of the
 * dense layer, fdgan_bn_bwd_finalize_coef's B and C) is still pending: the kernel uses dy + dy_affine_b[c] * dy_affine_x + dy_affine_c[c]
 * (dy_affine_x: that norm's input, the 128-channel bottleneck activation; rounded to bf16 as fdgan_affine_accumulate would have
 * stored it), which replaces that read-read-write pass over the gradient buffer.  dy itself is left as it is.
 * dw == NULL: the kernel's per-slot partials stay in wgrad_workspace ([*wsplit_out][128][C] fp32) and the caller sums them
 * later -- fdgan_wgrad_reduce_batch does that for every such conv of a backward walk in ONE launch, off the chain of
 * dependent launches (the weight gradient only feeds the optimizer). */
int fdgan_conv1x1_bwd_data_weight(const FdTensor* dy, const void* w_packed_flipped, const FdTensor* fwd_x, const FdPrologue* fwd_pro,
                                  const FdTensor* dpre, int accumulate, float* partial, int64_t capacity_floats, int64_t* rows_out,
                                  int64_t* cpad_out, float* wgrad_workspace, int64_t wgrad_workspace_floats, float* dw, int dw_accumulate,
                                  const FdTensor* dy_affine_x, const float* dy_affine_b, const float* dy_affine_c, int64_t* wsplit_out, FdStream stream);
/* bsum[c] += B, csum[c] += C of dx = A*dpre + B*x + C for channels [0, channels): B = -gamma*rstd^2*dgamma/count,
 * C = -gamma*rstd*dbeta/count - B*mean (pro: the forward prologue's mean / var / gamma / eps). */
int fdgan_bn_bwd_coef(const float* dgamma, const float* dbeta, const FdPrologue* pro, int64_t channels, int64_t count,
                      float* bsum, float* csum, FdStream stream);
/* Sums [nsplit][numel] fp32 partial blocks into out[numel] (+= when accumulate) for a whole table of jobs in one launch, in
 * the fixed order of the per-conv reduction (bitwise the same sums).  `jobs` lives in DEVICE memory; first_group = running
 * sum of ceil(numel / 64) over the preceding jobs, total_groups = that sum over all jobs. */
/* Zero fills as launches of this library, so that they can be recorded into a plan (torch's `t.zero_()` cannot): `rows` rows of
 * `row_bytes` bytes, `row_stride_bytes` apart (4-byte granularity); and a whole device table of buffers in one launch
 * (16-byte-aligned pointers and sizes; first_group = running sum of ceil(bytes / 16384) over the preceding jobs) -- the gradient
 * buffers a reverse walk accumulates into.  fdgan_add_transposed_f32: dst[c][r] += src[r][c] (dst is cols x rows): the
 * gradient of a ConvTranspose2d 1x1 filter, stored (cin, cout), from the (cout, cin) matrix the weight-gradient kernels write
 * (/root/reference/models/dehaze1113.py:362-364). */
typedef struct FdZeroJob {
  void* ptr;
  int64_t bytes;
  int64_t first_group;
} FdZeroJob;
int fdgan_fill_zero(void* p, int64_t row_bytes, int64_t rows, int64_t row_stride_bytes, FdStream stream);
int fdgan_fill_zero_many(const FdZeroJob* jobs_device, int64_t njobs, int64_t total_groups, FdStream stream);
int fdgan_add_transposed_f32(float* dst, const float* src, int64_t rows, int64_t cols, FdStream stream);

typedef struct FdReduceJob {
  const float* part;
  float* out;
  int64_t numel;
  int32_t nsplit;
  int32_t accumulate;
  int64_t first_group;
} FdReduceJob;
int fdgan_wgrad_reduce_batch(const FdReduceJob* jobs_device, int64_t njobs, int64_t total_groups, FdStream stream);
/* The same for the row-walking weight-gradient kernels (3x3 / 4x4 stride 1: the growth conv, the dy blocks, the discriminators),
 * whose partial sums are kept in MFMA accumulator order: fdgan_conv2d_bwd_weight_job is fdgan_conv2d_bwd_weight that also
 * DESCRIBES its final reduction in *job (job->part == NULL when the kernel chosen for the shape has no such reduction, or
 * produces a bias gradient: then everything has been launched as usual) and, with defer != 0, does not launch it: `workspace`
 * then holds the partial sums until the caller has run fdgan_wgrad_tr_reduce_batch over a device table of such jobs
 * (first_group = running sum of `groups` over the preceding jobs; everything else as the library filled it).  Same summation
 * order as the per-conv launch: bitwise the same gradients.  Replaces, per training step, 55 reduction launches by one
 * (torch.autograd's per-conv weight gradients at /root/reference/models/dehaze1113.py:711-724, :200-207). */
typedef struct FdTrReduceJob {
  const float* part;
  float* out;
  int64_t item_stride;
  int32_t items, zt, kyg, kyn, ks, nw, taps, cin, cout, accumulate;
  int64_t first_group;
  int64_t groups;
} FdTrReduceJob;
int fdgan_conv2d_bwd_weight_job(const FdTensor* x, const FdPrologue* pro, const FdTensor* dy, const FdConvDesc* d,
                                float* dw, float* dbias, float* workspace, int64_t workspace_floats, int accumulate,
                                FdTrReduceJob* job, int defer, FdStream stream);
int fdgan_wgrad_tr_reduce_batch(const FdTrReduceJob* jobs_device, int64_t njobs, int64_t total_groups, FdStream stream);

/* dx += bsum[c] * x + csum[c] (x: NHWC fp16 activation, dx: NHWC bf16 gradient of equal shape). */
int fdgan_affine_accumulate(const FdTensor* x, const float* bsum, const float* csum, const FdTensor* dx, FdStream stream);
/* ABI v16: the same sum written to ANOTHER view, out = g + bsum * x + csum (g is not modified; same rounding, same random bits as the
 * in-place form).  The backward walk uses it to hand the flushed gradient of a 32-channel growth slice -- 64-byte pieces at the concat
 * buffer's pitch -- to its two readers as a pixel-dense tensor (half the 128-byte lines per read). */
int fdgan_affine_accumulate_out(const FdTensor* x, const float* bsum, const float* csum, const FdTensor* g, const FdTensor* out, FdStream stream);
/* Pooled prologues (pro->pool2, the transitions: BatchNorm + ReLU + 2x2 average in front of the 1x1 conv,
 * torchvision _Transition as used at dehaze1113.py:716-728): fdgan_bn_act_bwd and fdgan_bn_bwd_apply accept the HALF-resolution
 * gradient w.r.t. the pooled activation as `da` / `dpre` and un-pool (x 1/4) and mask on the fly -- bn_act_bwd then only
 * produces the sums (nothing is written back), bn_bwd_apply forms dpre per pixel.  No full-resolution dpre tensor exists. */
int fdgan_bn_bwd_apply(const FdTensor* dpre, const FdTensor* x, const FdPrologue* pro, const float* dgamma,
                       const float* dbeta, const FdTensor* dx, int accumulate, FdStream stream);
int fdgan_conv2d_bwd_data_direct(const FdTensor* dy, const float* w, int cout, int cin, const FdConvDesc* d,
                                 float* dx, int64_t n, int64_t h, int64_t wd, FdStream stream);
int fdgan_conv2d_bwd_data_direct_nhwc(const FdTensor* dy, const float* w, int cout, int cin, const FdConvDesc* d,
                                      const FdTensor* dx, FdStream stream);
int fdgan_out_act_bwd(const float* dout, const float* out, int64_t n, int64_t c, int64_t h, int64_t w, int act,
                      const FdTensor* g, FdStream stream);
int fdgan_grad_ew(int mode, const FdTensor* src, const FdTensor* ref, const FdTensor* dst, FdStream stream);

/* torch.optim.Adam's update (no weight decay, no amsgrad -- what the reference's lrG / lrD / beta1 flags configure,
 * demo.py:43-46) on one flat fp32 tensor: exp_avg m, exp_avg_sq v, 1-based `step` for the bias corrections. */
int fdgan_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int64_t step, FdStream stream);

/* Differentiable SSIM (models/pytorch_ssim/__init__.py:8-73: 11x11 Gaussian window sigma 1.5, depthwise, zero
 * padding 5, C1 = 0.01^2, C2 = 0.03^2) on fp32 NCHW planes.  fdgan_ssim_fwd writes per-tile partial sums of the
 * SSIM map (mean = sum / (planes*h*w)) and three planes of partial derivatives; fdgan_ssim_bwd turns them into
 * d mean / d x, scaled by `weight`. */
int fdgan_ssim_fwd(const float* x, const float* y, int64_t planes, int64_t h, int64_t w, float* partial,
                   int64_t partial_floats, float* da, float* db, float* dc, FdStream stream);
int fdgan_ssim_bwd(const float* x, const float* y, const float* da, const float* db, const float* dc, int64_t planes,
                   int64_t h, int64_t w, float weight, float* dx, FdStream stream);

/* SSIM(window_size) for odd window sizes <= 11 (models/pytorch_ssim/__init__.py:39-73 takes the argument; 11 is the default the
 * reference uses): the zero-extended window on the same kernels. */
int fdgan_ssim_fwd_w(const float* x, const float* y, int64_t planes, int64_t h, int64_t w, int window_size, float* partial,
                     int64_t partial_floats, float* da, float* db, float* dc, FdStream stream);
int fdgan_ssim_bwd_w(const float* x, const float* y, const float* da, const float* db, const float* dc, int64_t planes,
                     int64_t h, int64_t w, int window_size, float weight, float* dx, FdStream stream);

/* ---- scalar losses -----------------------------------------------------------------------------------
 * Mean reductions of a training loop over the reference's networks, value AND gradient in one pass, ordered two-stage
 * sums (bit-reproducible).  fdgan_loss_f32 on contiguous fp32 tensors of n elements:
 *   kind 0  F.l1_loss(x, t)                  grad = sign(x - t) / n
 *   kind 1  F.mse_loss(x, t)                 grad = 2 (x - t) / n
 *   kind 2  F.binary_cross_entropy(x, t)     logs clamped at -100, grad = (x - t) / max(x (1 - x), 1e-12) / n  (torch's
 *           rule) -- D's sigmoid map, dehaze1113.py:221-223, against all-ones / all-zeros
 * t == NULL: the constant target t_const.  `partial` receives *nparts per-workgroup sums of the UNSCALED loss terms;
 * fdgan_sum_partials(partial, nparts, 1.0 / n, out) turns them into the mean (fp64 accumulation, index order).
 * grad (optional): d mean / d x, fp32, same shape.
 * fdgan_mse_nhwc_fwd / _bwd: mean squared difference of two NHWC fp16 views (Vgg16's tapped feature maps; the gradient view is bf16,
 * myutils/vgg16.py:27-49, read where the conv kernels left them): partial sums pre-multiplied by `scale` (1 / numel for
 * a mean; several maps may share one partial array and one fdgan_sum_partials), and g = upstream[0] * scale * (a - b)
 * written to the gradient view of `a` (scale = 2 / numel), `upstream` a DEVICE scalar so no host sync is needed.
 * relu_mask != 0: `a` is the stored output of a ReLU epilogue and g is zeroed where a == 0, i.e. the gradient is already
 * the one w.r.t. the PRE-activation (vgg16.py:28-46: every tapped map is F.relu(conv)), which lets the backward walk skip
 * its separate mask pass. */
int fdgan_loss_f32(int kind, const float* x, const float* t, float t_const, int64_t n, float* grad, float* partial,
                   int64_t partial_floats, int64_t* nparts, FdStream stream);
int fdgan_sum_partials(const float* partial, int64_t count, double scale, float* out, FdStream stream);
int fdgan_mse_nhwc_fwd(const FdTensor* a, const FdTensor* b, float scale, float* partial, int64_t partial_floats,
                       int64_t* nparts, FdStream stream);
int fdgan_mse_nhwc_bwd(const FdTensor* a, const FdTensor* b, const float* upstream, float scale, int relu_mask,
                       const FdTensor* g, FdStream stream);

/* ContextualLoss (the reference's bytecode-only loss module, original loss.py:23-73): the row part.
 * d: cosine distances [rows = B*HW][n = HW] fp32 contiguous.  relative_distances -> weighted_average_distances -> max over
 * j collapse to m_i = 1 / sum_j exp((dmin_i - d_ij) / (sigma (dmin_i + eps))) (the constant b cancels): one fused pass per
 * row instead of three HW x HW intermediates.  fwd writes m and the per-row (dmin, S, first argmin) the backward needs;
 * bwd writes gd = dL/dd from gm = dL/dm. */
int fdgan_cx_rows_fwd(const float* d, int64_t rows, int64_t n, float sigma, float eps, float* m, float* dmin, float* S,
                      int32_t* jmin, FdStream stream);
int fdgan_cx_rows_bwd(const float* d, int64_t rows, int64_t n, float sigma, float eps, const float* dmin, const float* S,
                      const int32_t* jmin, const float* gm, float* gd, FdStream stream);

/* F.max_pool2d(h, kernel_size=2, stride=2) (myutils/vgg16.py:31,36,42) on NHWC fp16 views (backward: x fp16, dy / dx bf16);
 * y is (n, h/2, w/2, c), c a multiple of 8. */
int fdgan_maxpool2_nhwc(const FdTensor* x, const FdTensor* y, FdStream stream);
/* its backward: dx += dy routed to the first maximum of each 2x2 window of x (F.max_pool2d's tie rule) */
int fdgan_maxpool2_bwd_nhwc(const FdTensor* x, const FdTensor* dy, const FdTensor* dx, FdStream stream);

/* ---- frequency split of the Fusion-discriminator input ------------------------- */
/* Reference: __pycache__/loss.cpython-36.pyc (source loss.py absent).
 * x, y: NCHW fp32 contiguous, (n, c, h, w).
 * fdgan_blur15_fwd     = Blur(l=15, isotropic_gaussian_kernel(15, 3.0)).forward  (loss.py:122-162):
 *                        optional (x - ImageNet mean) / std (c must be 3), ReflectionPad2d(7),
 *                        the same normalised 15x15 Gaussian on every (b, c) plane.  h, w >= 8.
 * fdgan_laplacian3_fwd = Laplacian(3).forward (loss.py:245-304): depthwise 3x3, ones with
 *                        centre -8, zero padding 1, not normalised.
 * fdgan_fusion_input_nhwc fuses both with the layout change D needs: reads img once and writes
 *   channels [img | Blur(img) | Laplacian(img)] (3*c) of the NHWC fp16 view y (the concat fed to
 *   D, facades/network.png); channels of y beyond 3*c are left untouched. */
int fdgan_blur15_fwd(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, int use_input_norm,
                     FdStream stream);
int fdgan_laplacian3_fwd(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, FdStream stream);
/* Laplacian(kernel_size) for any odd kernel_size in 3 .. 15 (the class of loss.py:245-301 takes one; the network builds
 * Laplacian(3), loss.py:304): k x k box sum with zero padding (k - 1) / 2 minus k^2 x the centre, depthwise, unnormalised.
 * Self-adjoint: its backward is the same call on dy. */
int fdgan_laplacian_fwd(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, int ksize, FdStream stream);
/* What the Fusion-discriminator is fed (train.py's fusion_input; /root/reference/facades/network.png, the frequency split of
 * /root/reference/__pycache__/loss.cpython-36.pyc): out[n] = cat([img[n], Blur(img)[n], Laplacian(img)[n]]) as NCHW fp32,
 * (n, 3 c, h, w) in ONE buffer -- the two filters write their planes where the concatenation wants them and the Laplacian
 * pass, which holds every input row in registers anyway, stores the image planes too: 1 read + 3 writes of the image instead
 * of the filters' 2 reads + 2 writes followed by a 3-plane copy.  Needs W % 4 == 0, W and H >= 16 and 16-byte aligned tensors:
 * FD_EUNSUPPORTED otherwise (nothing launched; use the two filters and a concatenation). */
int fdgan_fusion_input_nchw(const float* img, float* out, int64_t n, int64_t c, int64_t h, int64_t w, int use_input_norm,
                            FdStream stream);
/* Backward of the two filters (they feed D, whose gradient reaches the generator through them): the Laplacian
 * is self-adjoint (symmetric kernel, zero padding): call fdgan_laplacian3_fwd on dy.  Blur's adjoint folds the
 * reflection halo back; `tmp` is n*c*h*w floats of caller-owned scratch. */
int fdgan_blur15_bwd(const float* dy, float* tmp, float* dx, int64_t n, int64_t c, int64_t h, int64_t w,
                     int use_input_norm, FdStream stream);
/* Blur(l, isotropic_gaussian_kernel(l, sigma)) for any odd l <= 15 and sigma > 0 (the module's constructor arguments,
 * loss.py:122-159; the reference builds l = 15, sigma = 3): the 15-tap kernels on zero-extended taps.  h, w >= 8 (16 backward). */
int fdgan_blur_gauss_fwd(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, int l, float sigma,
                         int use_input_norm, FdStream stream);
int fdgan_blur_gauss_bwd(const float* dy, float* tmp, float* dx, int64_t n, int64_t c, int64_t h, int64_t w, int l, float sigma,
                         int use_input_norm, FdStream stream);
/* dx = Laplacian(dy): the operator is self-adjoint (symmetric kernel, zero padding); loss.py:286-301 under autograd. */
int fdgan_laplacian3_bwd(const float* dy, float* dx, int64_t n, int64_t c, int64_t h, int64_t w, FdStream stream);
int fdgan_fusion_input_nhwc(const float* img, int64_t n, int64_t c, int64_t h, int64_t w, const FdTensor* y,
                            int use_input_norm, FdStream stream);

/* ---- measurement aid (tools/conv_bench.py) ------------------------------------- */
/* Registers a device buffer (>= 64 x int64) that instrumented kernel builds fill with per-wave
 * s_memtime phase totals of workgroup 0 (the round-1 phase analysis of the persistent 3x3 kernel
 * in DESIGN.md was taken this way).  The shipped kernels carry no instrumentation; NULL disables. */
int fdgan_debug_timing(void* device_buf);

/* Kernel timer (measurement aid, used by bench.py's roofline leg): while armed, every `stride`-th launch whose launcher
 * name equals `name` (NULL or "*": every launch of the library, eager or replayed from a plan) is bracketed by a
 * hipEvent pair on the stream it is launched on, up to max_samples.  fdgan_kernel_timer_read waits for the events,
 * returns per sample the index of the launch among the matching launches, its duration in ms and (names48: 48 bytes
 * per sample, may be NULL) its launcher name, and disarms.  One timer per process; not thread-safe. */
int fdgan_kernel_timer_arm(const char* name, int stride, int max_samples);
int fdgan_kernel_timer_read(int capacity, int* n_out, int* matching_launches_out, int* call_index, float* ms, char* names48);

/* ---- data-parallel gradient exchange for non-PyTorch hosts (csrc/allreduce.hip) ----------------------------------
 * One process per GPU.  Rank 0 obtains the 128-byte id and hands it to the other ranks out of band (file, TCP, an
 * existing store); every rank then creates its communicator (collective: all `world` ranks must call), sums its
 * gradient buffer IN PLACE on its own stream (asynchronous; divide by `world` afterwards, e.g. in the optimizer step)
 * and destroys the communicator at exit.  FD_EUNSUPPORTED when librccl.so cannot be loaded. */
#define FDGAN_ALLREDUCE_ID_BYTES 128
int fdgan_allreduce_unique_id(void* id128);
int fdgan_allreduce_comm_create(const void* id128, int rank, int world, void** comm);
int fdgan_allreduce_sum_f32(void* comm, float* buf, int64_t count, FdStream stream);
int fdgan_allreduce_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* FDGAN_HIP_H */
