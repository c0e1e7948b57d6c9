// conv_igemm.hip -- argument validation and the C-ABI entry points of the fused convolution.
// Kernel: conv_igemm.h; instantiations: conv_k1.hip, conv_k3.hip, conv_k4.hip.
#include <stdlib.h>

#include "conv_igemm.h"

unsigned long long* g_fd_debug_timing = nullptr;
extern "C" int fdgan_debug_timing(void* device_buf) {
  g_fd_debug_timing = static_cast<unsigned long long*>(device_buf);
  return FD_OK;
}

extern "C" int fdgan_conv_weight_layout(int cout, int cin, int ksize, int stride) {
  // 1x1 stride-1 filters that fit LDS run on the x-stream kernel (conv1x1_xs.hip), which
  // consumes the "x64" fragment order; everything else uses the 32-channel chunk order.
  if (ksize == 1 && stride == 1 && cout > 0 && cin > 0 && conv1x1_xs_fits(cout, cin)) return FD_WLAYOUT_X64;
  return FD_WLAYOUT_CHUNK32;
}

static int conv_dispatch(ConvArgs& a, long long nimg, int cout_total, int ksize, int stride, bool pool,
                         int w_layout, FdConvInfo* info, long long stats_cap, bool dry, hipStream_t stream) {
  if (a.grad_io && (w_layout != FD_WLAYOUT_CHUNK32 || stride != 1 || pool || (a.pro_mode != 0 && a.mk_mode == 0)))
    FD_FAIL(FD_EUNSUPPORTED, "conv2d on bf16 gradients: stride-1, chunk32 filter image, no prologue");
  if (w_layout == FD_WLAYOUT_X64) {
    if (ksize != 1 || stride != 1) FD_FAIL(FD_EINVAL, "conv2d: the x64 weight layout is for 1x1 stride-1 convs");
    return conv_dispatch_k1_xs(a, nimg, cout_total, pool, info, stats_cap, dry, stream);
  }
  if (w_layout != FD_WLAYOUT_CHUNK32) FD_FAIL(FD_EINVAL, "conv2d: unknown weight layout %d", w_layout);
  if (conv_cout1_fits(a, cout_total, ksize, stride, pool)) return conv_cout1_launch(a, nimg, ksize, info, dry, stream);
  if (const int sc = conv_sc_variant(a, cout_total, ksize, stride, pool)) return conv_sc_launch(sc, a, nimg, info, stats_cap, dry, stream);
  switch (ksize) {
    case 1: return conv_dispatch_k1(a, nimg, cout_total, stride, pool, info, stats_cap, dry, stream);
    case 3: return conv_dispatch_k3(a, nimg, cout_total, stride, pool, info, stats_cap, dry, stream);
    case 4: return conv_dispatch_k4(a, nimg, cout_total, stride, pool, info, stats_cap, dry, stream);
  }
  FD_FAIL(FD_EUNSUPPORTED, "no kernel for ksize=%d stride=%d", ksize, stride);
}

static int conv_setup(const FdTensor* x, const void* w_packed, const float* bias, const FdPrologue* pro,
                      const FdTensor* y, int cout, const FdStats* stats, const FdConvDesc* d, ConvArgs& a,
                      long long& nimg, bool& pool) {
  FD_REQUIRE(x && y && d, "conv2d: NULL tensor/descriptor");
  FD_REQUIRE(x->dtype == FD_F16 || x->dtype == FD_BF16, "conv2d: x must be an NHWC fp16 (activation) or bf16 (gradient) view");
  FD_REQUIRE(x->stride[3] == 1 && x->stride[2] % 8 == 0 && x->stride[1] % 8 == 0 && x->stride[0] % 8 == 0,
             "conv2d: x strides must be channel-contiguous and multiples of 8 elements");
  FD_REQUIRE(((uintptr_t)x->ptr & 15) == 0, "conv2d: x pointer must be 16-byte aligned");
  FD_REQUIRE(d->ksize == 1 || d->ksize == 3 || d->ksize == 4, "conv2d: ksize %d", d->ksize);
  FD_REQUIRE(d->stride == 1 || d->stride == 2, "conv2d: stride %d", d->stride);
  pool = pro && pro->pool2;
  const long long hs = x->h, ws = x->w;
  const long long hin = pool ? hs / 2 : hs, win = pool ? ws / 2 : ws;
  const long long ho = (hin + 2 * d->pad - d->ksize) / d->stride + 1;
  const long long wo = (win + 2 * d->pad - d->ksize) / d->stride + 1;
  FD_REQUIRE(ho > 0 && wo > 0, "conv2d: empty output");
  const int up = d->upsample2 ? 2 : 1;
  FD_REQUIRE(y->n == x->n && y->h == ho * up && y->w == wo * up,
             "conv2d: y is %lldx%lldx%lld, expected %lldx%lldx%lld", (long long)y->n, (long long)y->h,
             (long long)y->w, (long long)x->n, ho * up, wo * up);
  FD_REQUIRE(x->stride[1] * hs < (1ll << 31) && y->stride[1] * y->h < (1ll << 31),
             "conv2d: one image exceeds 2^31 elements");
  FD_REQUIRE(x->stride[1] < (1ll << 31) && x->stride[2] < (1ll << 31), "conv2d: stride overflow");
  a = ConvArgs{};
  a.grad_io = x->dtype == FD_BF16;   // a convolution over gradients (data gradient): bf16 in, bf16 filter image, bf16 out
  a.x = static_cast<const unsigned short*>(x->ptr);
  a.x_sn = x->stride[0];
  a.x_sh = (int)x->stride[1];
  a.x_sw = (int)x->stride[2];
  a.Hs = (int)hs;
  a.Ws = (int)ws;
  a.Cin = (int)x->c;
  a.Cin8 = (int)((x->c + 7) / 8);
  a.nchunk = (int)((x->c + 31) / 32);
  FD_REQUIRE(x->stride[2] >= a.Cin8 * 8 || x->w == 1, "conv2d: pixel pitch %lld < padded Cin %d",
             (long long)x->stride[2], a.Cin8 * 8);
  a.w = static_cast<const unsigned short*>(w_packed);
  a.ntile_total = (cout + 15) / 16;
  a.CoutW = cout;
  a.bias = bias;
  a.pro_mode = 0;
  a.p_slope = 1.f;
  if (pro) {
    FD_REQUIRE(pro->act == FD_ACT_NONE || pro->act == FD_ACT_RELU || pro->act == FD_ACT_LEAKY02,
               "conv2d: prologue activation %d", pro->act);
    a.p_slope = pro->act == FD_ACT_RELU ? 0.f : (pro->act == FD_ACT_LEAKY02 ? 0.2f : 1.f);
    if (pro->mean) {
      FD_REQUIRE(pro->var, "conv2d: prologue mean without var");
      a.pro_mode = 2;
      a.p_mean = pro->mean;
      a.p_var = pro->var;
      a.p_gamma = pro->gamma;
      a.p_beta = pro->beta;
      a.eps = pro->eps;
      a.momentum = pro->momentum;
      a.run_mean = pro->running_mean;
      a.run_var = pro->running_var;
      a.nbt = reinterpret_cast<long long*>(pro->num_batches_tracked);
      FD_REQUIRE(!pro->running_mean || pro->running_var, "conv2d: running_mean without running_var");
      a.unbias = pro->count > 1 ? (float)((double)pro->count / (double)(pro->count - 1)) : 1.f;
    } else if (pro->act != FD_ACT_NONE || pool) {
      a.pro_mode = 1;
    }
  }
  a.y = y->ptr;
  a.Ho = (int)ho;
  a.Wo = (int)wo;
  a.Cout = (int)y->c;
  FD_REQUIRE(y->c <= ((cout + 15) / 16) * 16 && y->c >= 1, "conv2d: y->c=%lld vs cout=%d", (long long)y->c, cout);
  FD_REQUIRE(d->epilogue_act >= FD_ACT_NONE && d->epilogue_act <= FD_ACT_SIGMOID, "conv2d: epilogue activation %d",
             d->epilogue_act);
  // ReLU / LeakyReLU are fused as max(v, slope*v); tanh / sigmoid run as a second, elementwise
  // launch over what the conv stored (they only follow the two tiny final convolutions)
  a.e_slope = d->epilogue_act == FD_ACT_RELU ? 0.f : (d->epilogue_act == FD_ACT_LEAKY02 ? 0.2f : 1.f);
  if (d->epilogue_act == FD_ACT_TANH || d->epilogue_act == FD_ACT_SIGMOID)
    FD_REQUIRE(!stats, "conv2d: batch statistics are not available after a tanh/sigmoid epilogue");
  a.upsample = d->upsample2 ? 1 : 0;
  a.pad = d->pad;
  if (y->dtype == FD_F32) {
    // NCHW fp32: strides given as n, h, w, c element strides
    a.out_nchw_f32 = 1;
    a.y_sn = y->stride[0];
    a.y_sh = (int)y->stride[1];
    a.y_sw = (int)y->stride[2];
    a.y_sc = y->stride[3];
  } else {
    FD_REQUIRE(y->dtype == x->dtype, "conv2d: a 16-bit y must have x's element format (fp16 activations / bf16 gradients), got %d vs %d",
               y->dtype, x->dtype);
    FD_REQUIRE(y->stride[3] == 1 && y->stride[2] % 4 == 0 && y->stride[1] % 4 == 0 && y->stride[0] % 4 == 0,
               "conv2d: y strides must be channel-contiguous, multiples of 4 elements");
    FD_REQUIRE(((uintptr_t)y->ptr & 7) == 0, "conv2d: y pointer must be 8-byte aligned");
    a.out_nchw_f32 = 0;
    a.y_vec16 = !d->upsample2 && ((uintptr_t)y->ptr & 15) == 0 && y->stride[2] % 8 == 0 && y->stride[1] % 8 == 0 &&
                y->stride[0] % 8 == 0;
    a.y_sn = y->stride[0];
    a.y_sh = (int)y->stride[1];
    a.y_sw = (int)y->stride[2];
    a.y_sc = 1;
  }
  a.stats = stats ? stats->partial : nullptr;
  if (stats && stats->mean) {
    FD_REQUIRE(stats->var && stats->counter && stats->count > 0, "conv2d: fused finalize needs var, counter and count");
    a.fin_mean = stats->mean;
    a.fin_var = stats->var;
    a.fin_counter = stats->counter;
    a.fin_inv_count = 1.0 / (double)stats->count;
  }
  a.dbg = g_fd_debug_timing;
  {  // measurement aid (tools only): FDGAN_DEBUG_NOSTORE=1 drops every output store
    static const bool nostore = FD_TUNE_GETENV("FDGAN_DEBUG_NOSTORE") != nullptr;
    if (nostore) a.Cout = 0;
    static const char* co = FD_TUNE_GETENV("FDGAN_DEBUG_COALESCE");
    if (co && d->ksize == 1) a.pad = atoi(co);
    static const char* ph = FD_TUNE_GETENV("FDGAN_DEBUG_PHASES");   // bit mask of kernel phases to skip (results wrong)
    a.dbg_skip = ph ? atoi(ph) : 0;
  }
  nimg = x->n;
  return FD_OK;
}

extern "C" int fdgan_conv2d_fwd_info(const FdTensor* x, const FdTensor* y, int cout, const FdConvDesc* d,
                                     const FdPrologue* pro, FdConvInfo* info) {
  FD_REQUIRE(info, "conv2d_fwd_info: NULL info");
  ConvArgs a;
  long long nimg;
  bool pool;
  int rc = conv_setup(x, nullptr, nullptr, pro, y, cout, nullptr, d, a, nimg, pool);
  if (rc != FD_OK) return rc;
  return conv_dispatch(a, nimg, cout, d->ksize, d->stride, pool, d->w_layout, info, -1, true, nullptr);
}

extern "C" int fdgan_conv2d_fwd(const FdTensor* x, const void* w_packed, const float* bias,
                                const FdPrologue* pro, const FdTensor* y, const FdStats* stats,
                                const FdConvDesc* d, FdStream stream) {
  FD_REQUIRE(w_packed, "conv2d_fwd: NULL weights");
  FD_REQUIRE(((uintptr_t)w_packed & 15) == 0, "conv2d_fwd: packed weights must be 16-byte aligned");
  ConvArgs a;
  long long nimg;
  bool pool;
  FD_REQUIRE(d && y, "conv2d_fwd: NULL descriptor/output");
  const int cout = d->cout > 0 ? d->cout : (int)y->c;
  int rc = conv_setup(x, w_packed, bias, pro, y, cout, stats, d, a, nimg, pool);
  if (rc != FD_OK) return rc;
  if (a.grad_io) {
    // a plain convolution over bf16 gradients (the unfused data gradient: dy * flip(W)): runs on the backward-data
    // instantiations with the mask switched off; their epilogue moves whole 16-byte pieces of pixel rows
    FD_REQUIRE(!bias && !stats && !pro && d->epilogue_act == FD_ACT_NONE && !d->upsample2 && y->dtype == FD_BF16 && a.y_vec16 &&
                   (y->stride[2] >= (y->c + 7) / 8 * 8 || y->w == 1),
               "conv2d_fwd on bf16 gradients: no bias / prologue / epilogue / statistics; y rows 16-byte aligned, channels padded to 8");
  }
  rc = conv_dispatch(a, nimg, cout, d->ksize, d->stride, pool, d->w_layout, nullptr,
                     stats ? stats->capacity_floats : -1, false, static_cast<hipStream_t>(stream));
  if (rc != FD_OK) return rc;
  if (d->epilogue_act == FD_ACT_TANH || d->epilogue_act == FD_ACT_SIGMOID)
    return fd_act_inplace(y, d->epilogue_act, static_cast<hipStream_t>(stream));
  return FD_OK;
}

/* Data AND weight gradient of the dense-layer bottleneck (1x1, 128 filters) in one pass over dy and x (include/fdgan_hip.h). */
extern "C" int fdgan_conv1x1_bwd_data_weight(const FdTensor* dy, const void* w_packed_flipped, const FdTensor* fwd_x, const FdPrologue* fwd_pro,
                                             const FdTensor* dpre, int accumulate, float* partial, int64_t capacity_floats, int64_t* rows_out,
                                             int64_t* cpad_out, float* wgrad_workspace, int64_t wgrad_workspace_floats, float* dw,
                                             int dw_accumulate, const FdTensor* dy_affine_x, const float* dy_affine_b, const float* dy_affine_c,
                                             int64_t* wsplit_out, FdStream stream) {
  FD_REQUIRE(dy && w_packed_flipped && fwd_x && dpre && wgrad_workspace, "conv1x1_bwd_data_weight: NULL argument");
  FD_REQUIRE(accumulate >= 0 && accumulate <= 2, "conv1x1_bwd_data_weight: accumulate %d", accumulate);
  FD_REQUIRE(((uintptr_t)w_packed_flipped & 15) == 0, "conv1x1_bwd_data_weight: packed weights must be 16-byte aligned");
  FD_REQUIRE(fwd_pro == nullptr || !fwd_pro->pool2, "conv1x1_bwd_data_weight: pooled prologues are not fused");
  const int act0 = fwd_pro ? fwd_pro->act : FD_ACT_NONE;
  FD_REQUIRE(act0 == FD_ACT_NONE || act0 == FD_ACT_RELU || act0 == FD_ACT_LEAKY02, "conv1x1_bwd_data_weight: prologue activation %d", act0);
  const bool fits = dy->dtype == FD_BF16 && dpre->dtype == FD_BF16 && fwd_x->dtype == FD_F16 && dy->c == 128 && dy->n == dpre->n &&
                    dy->h == dpre->h && dy->w == dpre->w && fwd_x->n == dpre->n && fwd_x->h == dpre->h && fwd_x->w == dpre->w &&
                    fwd_x->c >= dpre->c && conv1x1_bwd_fits(dy, fwd_x, dpre);
  if (!fits) FD_FAIL(FD_EUNSUPPORTED, "conv1x1_bwd_data_weight: shape outside the fused kernel (use fdgan_conv2d_bwd_data + fdgan_conv2d_bwd_weight)");
  FD_REQUIRE(!(fwd_pro && fwd_pro->mean) || (fwd_pro->var && partial), "conv1x1_bwd_data_weight: a BatchNorm prologue needs var and the partial-sum workspace");
  long long rows = 0, cpad = 0, nsplit = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dy_affine_x != nullptr) {
    auto dense = [](const FdTensor* t) {
      return t->stride[3] == 1 && t->stride[1] == t->w * t->stride[2] && t->stride[0] == t->h * t->stride[1] && t->stride[2] % 8 == 0 &&
             ((uintptr_t)t->ptr & 15) == 0;
    };
    FD_REQUIRE(dy_affine_b && dy_affine_c, "conv1x1_bwd_data_weight: dy_affine_x without its coefficients");
    FD_REQUIRE(dy_affine_x->dtype == FD_F16 && dy_affine_x->c == 128 && dy_affine_x->n == dy->n && dy_affine_x->h == dy->h &&
                   dy_affine_x->w == dy->w && dense(dy_affine_x),
               "conv1x1_bwd_data_weight: dy_affine_x must be a dense 128-channel NHWC fp16 view shaped like dy");
  }
  const int rc = conv1x1_bwd_launch(dy, w_packed_flipped, fwd_x, fwd_pro, dpre, accumulate, partial, capacity_floats, &rows, &cpad, st,
                                    wgrad_workspace, wgrad_workspace_floats, &nsplit, dy_affine_x, dy_affine_b, dy_affine_c);
  if (rc == 1) FD_FAIL(FD_EUNSUPPORTED, "conv1x1_bwd_data_weight: the weight-gradient workspace cannot hold one partial per pixel slot");
  if (rc != FD_OK) return rc;
  if (rows_out) *rows_out = rows;
  if (cpad_out) *cpad_out = cpad;
  if (wsplit_out) *wsplit_out = nsplit;
  if (dw == nullptr) return FD_OK;      // the caller reduces the partials later (fdgan_wgrad_reduce_batch)
  return fd_wgrad_reduce(wgrad_workspace, dw, 128LL * dpre->c, (int)nsplit, dw_accumulate, st);
}

/* Data gradient of a stride-1 conv fused with the backward of the conv's input-side prologue (include/fdgan_hip.h). */
extern "C" int fdgan_conv2d_bwd_data(const FdTensor* dy, const void* w_packed_flipped, const FdTensor* fwd_x,
                                     const FdPrologue* fwd_pro, const FdTensor* dpre, int accumulate, float* partial,
                                     int64_t capacity_floats, int64_t* rows_out, int64_t* cpad_out, const FdConvDesc* d,
                                     FdStream stream) {
  FD_REQUIRE(w_packed_flipped && d && dpre && fwd_x, "conv2d_bwd_data: NULL argument");
  FD_REQUIRE(accumulate >= 0 && accumulate <= 2, "conv2d_bwd_data: accumulate %d", accumulate);
  FD_REQUIRE(((uintptr_t)w_packed_flipped & 15) == 0, "conv2d_bwd_data: packed weights must be 16-byte aligned");
  FD_REQUIRE(d->stride == 1 && !d->upsample2 && d->epilogue_act == FD_ACT_NONE && d->w_layout == FD_WLAYOUT_CHUNK32,
             "conv2d_bwd_data: stride-1 conv with the chunk32 filter image, no epilogue");
  FD_REQUIRE(fwd_pro == nullptr || !fwd_pro->pool2, "conv2d_bwd_data: a pooled prologue is differentiated at full resolution");
  const int64_t c8 = (dpre->c + 7) / 8 * 8;
  FD_REQUIRE(dy->dtype == FD_BF16 && dpre->dtype == FD_BF16 && fwd_x->dtype == FD_F16 && fwd_x->n == dpre->n && fwd_x->h == dpre->h &&
                 fwd_x->w == dpre->w && fwd_x->c >= dpre->c && fwd_x->stride[3] == 1 && dpre->stride[3] == 1,
             "conv2d_bwd_data: dy / dpre are NHWC bf16 gradients, fwd_x an NHWC fp16 view shaped like dpre");
  for (const FdTensor* t : {fwd_x, dpre})   // the epilogue moves whole 16-byte pieces of pixel rows, pad channels included
    FD_REQUIRE(t->stride[2] % 8 == 0 && t->stride[1] % 8 == 0 && t->stride[0] % 8 == 0 && ((uintptr_t)t->ptr & 15) == 0 &&
                   (t->stride[2] >= c8 || t->w == 1),
               "conv2d_bwd_data: fwd_x / dpre need 16-byte aligned pixel rows with the channels padded to a multiple of 8");
  const int act0 = fwd_pro ? fwd_pro->act : FD_ACT_NONE;
  FD_REQUIRE(act0 == FD_ACT_NONE || act0 == FD_ACT_RELU || act0 == FD_ACT_LEAKY02, "conv2d_bwd_data: prologue activation %d", act0);
  {   // one forward filter behind an activation-only prologue (the discriminators' last conv): vector-ALU kernel, conv_c1.hip
    const int rc1 = dgrad_cout1_launch(dy, w_packed_flipped, fwd_x, fwd_pro, dpre, accumulate, d, static_cast<hipStream_t>(stream));
    if (rc1 != 1) {
      if (rows_out) *rows_out = 0;
      if (cpad_out) *cpad_out = 0;
      return rc1;
    }
  }
  if (d->ksize == 1 && d->pad == 0 && dy->dtype == FD_BF16 && dy->n == dpre->n && dy->h == dpre->h && dy->w == dpre->w &&
      (d->cout <= 0 || d->cout == dpre->c) && conv1x1_bwd_fits(dy, fwd_x, dpre)) {
    FD_REQUIRE(!(fwd_pro && fwd_pro->mean) || (fwd_pro->var && partial), "conv2d_bwd_data: a BatchNorm prologue needs var and the partial-sum workspace");
    long long rows = 0, cpad = 0;
    const int rc = conv1x1_bwd_launch(dy, w_packed_flipped, fwd_x, fwd_pro, dpre, accumulate, partial, capacity_floats, &rows, &cpad,
                                      static_cast<hipStream_t>(stream));
    if (rows_out) *rows_out = rows;
    if (cpad_out) *cpad_out = cpad;
    return rc;
  }
  if (dy->dtype == FD_BF16 && (d->cout <= 0 || d->cout == dpre->c) && conv3x3_bwd_fits(dy, fwd_x, dpre, d)) {
    FD_REQUIRE(!(fwd_pro && fwd_pro->mean) || (fwd_pro->var && partial), "conv2d_bwd_data: a BatchNorm prologue needs var and the partial-sum workspace");
    long long rows = 0, cpad = 0;
    const int rc = conv3x3_bwd_launch(dy, w_packed_flipped, fwd_x, fwd_pro, dpre, accumulate, partial, capacity_floats, &rows, &cpad,
                                      static_cast<hipStream_t>(stream));
    if (rows_out) *rows_out = rows;
    if (cpad_out) *cpad_out = cpad;
    return rc;
  }
  ConvArgs a;
  long long nimg;
  bool pool;
  const int cout = d->cout > 0 ? d->cout : (int)dpre->c;
  FdStats st{};
  st.partial = partial;
  st.capacity_floats = capacity_floats;
  const bool norm = fwd_pro && fwd_pro->mean;
  FD_REQUIRE(!norm || partial, "conv2d_bwd_data: a BatchNorm prologue needs the partial-sum workspace");
  int rc = conv_setup(dy, w_packed_flipped, nullptr, nullptr, dpre, cout, norm ? &st : nullptr, d, a, nimg, pool);
  if (rc != FD_OK) return rc;
  const int act = fwd_pro ? fwd_pro->act : FD_ACT_NONE;
  FD_REQUIRE(act == FD_ACT_NONE || act == FD_ACT_RELU || act == FD_ACT_LEAKY02, "conv2d_bwd_data: prologue activation %d", act);
  a.mk_mode = norm ? 2 : 1;
  a.mk_acc = accumulate;
  a.mk_slope = act == FD_ACT_RELU ? 0.f : (act == FD_ACT_LEAKY02 ? 0.2f : 1.f);
  a.mk_x = static_cast<const unsigned short*>(fwd_x->ptr);
  a.mk_sn = fwd_x->stride[0], a.mk_sh = (int)fwd_x->stride[1], a.mk_sw = (int)fwd_x->stride[2];
  if (norm) {
    FD_REQUIRE(fwd_pro->var, "conv2d_bwd_data: prologue mean without var");
    a.mk_mean = fwd_pro->mean, a.mk_var = fwd_pro->var, a.mk_gamma = fwd_pro->gamma, a.mk_beta = fwd_pro->beta, a.mk_eps = fwd_pro->eps;
  }
  FdConvInfo info{};
  rc = conv_dispatch(a, nimg, cout, d->ksize, d->stride, false, d->w_layout, &info, -1, true, nullptr);
  if (rc != FD_OK) return rc;
  if (rows_out) *rows_out = info.stats_rows;
  if (cpad_out) *cpad_out = info.stats_cpad;
  return conv_dispatch(a, nimg, cout, d->ksize, d->stride, false, d->w_layout, nullptr, norm ? capacity_floats : -1, false,
                       static_cast<hipStream_t>(stream));
}
