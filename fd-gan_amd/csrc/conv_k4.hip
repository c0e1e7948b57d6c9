// 4x4 convolutions of the discriminators (stride 2 and stride 1, dehaze1113.py:196,214,222).
#include "conv_igemm.h"

int conv_dispatch_k4(ConvArgs& a, long long nimg, int cout_total, int stride, bool pool, FdConvInfo* info,
                     long long stats_cap, bool dry, hipStream_t stream) {
  const bool narrow = cout_total <= 32, mid = cout_total <= 64;   // 32 / 64 / 128 output channels per workgroup
  if (pool) FD_FAIL(FD_EUNSUPPORTED, "pool2 prologue needs a 1x1 stride-1 conv");
  if (stride == 1 && a.mk_mode != 0) {   // backward data with the masked epilogue (fdgan_conv2d_bwd_data)
    if (narrow) FD_CONV_DISPATCH_X(4, 1, 0, 4, 2, 4, 1, 4, 1, "conv4x4_bn32_bwd");
    if (mid) FD_CONV_DISPATCH_X(4, 1, 0, 4, 4, 4, 1, 1, 1, "conv4x4_bn64_bwd");
    FD_CONV_DISPATCH_X(4, 1, 0, 4, 8, 4, 1, 1, 1, "conv4x4_bn128_bwd");
  }
  if (stride == 1) {
    if (narrow) FD_CONV_DISPATCH(4, 1, 0, 4, 2, 4, 1, 4, "conv4x4_bn32");
    if (mid) FD_CONV_DISPATCH(4, 1, 0, 4, 4, 4, 1, 1, "conv4x4_bn64");
    FD_CONV_DISPATCH(4, 1, 0, 4, 8, 4, 1, 1, "conv4x4_bn128");
  }
  if (stride == 2) {
    if (narrow) FD_CONV_DISPATCH(4, 2, 0, 2, 2, 4, 1, 4, "conv4x4s2_bn32");
    if (mid) FD_CONV_DISPATCH(4, 2, 0, 2, 4, 4, 1, 1, "conv4x4s2_bn64");
    FD_CONV_DISPATCH(4, 2, 0, 2, 8, 4, 1, 1, "conv4x4s2_bn128");
  }
  FD_FAIL(FD_EUNSUPPORTED, "4x4 conv with stride %d", stride);
}
