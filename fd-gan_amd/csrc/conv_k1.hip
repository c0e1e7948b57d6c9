// 1x1 convolutions (dense-layer bottlenecks, transitions with pooled prologue, decoder).
#include "conv_igemm.h"

int conv_dispatch_k1(ConvArgs& a, long long nimg, int cout_total, int stride, bool pool, FdConvInfo* info,
                     long long stats_cap, bool dry, hipStream_t stream) {
  const bool narrow = cout_total <= 32;
  if (stride != 1) FD_FAIL(FD_EUNSUPPORTED, "1x1 conv with stride %d", stride);
  if (!a.grad_io && pool) FD_CONV_DISPATCH(1, 1, 1, 2, 8, 4, 1, 1, "conv1x1_pool_bn128");
  if (a.grad_io) {   // backward data with the masked epilogue (fdgan_conv2d_bwd_data)
    if (pool) FD_FAIL(FD_EUNSUPPORTED, "masked backward-data epilogue with a pooled prologue");
    if (narrow) FD_CONV_DISPATCH_X(1, 1, 0, 4, 2, 4, 1, 1, 1, "conv1x1_bn32_bwd");
    FD_CONV_DISPATCH_X(1, 1, 0, 4, 8, 4, 1, 1, 1, "conv1x1_bn128_bwd");
  }
  if (narrow) FD_CONV_DISPATCH(1, 1, 0, 4, 2, 4, 1, 1, "conv1x1_bn32");
  FD_CONV_DISPATCH(1, 1, 0, 4, 8, 4, 1, 1, "conv1x1_bn128");
}
