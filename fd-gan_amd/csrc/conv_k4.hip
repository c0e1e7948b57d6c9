// 4x4 convolutions of the discriminators (stride 2 and stride 1, dehaze1113.py:196,214,222).
#include "conv_igemm.h"

int conv_dispatch_k4(ConvArgs& a, long long nimg, int cout_total, int stride, bool pool, FdConvInfo* info,
                     long long stats_cap, bool dry, hipStream_t stream) {
  const bool narrow = cout_total <= 32, mid = cout_total <= 64;   // 32 / 64 / 128 output channels per workgroup
  if (pool) FD_FAIL(FD_EUNSUPPORTED, "pool2 prologue needs a 1x1 stride-1 conv");
  if (stride == 1) {   // filter-direct kernels (conv_k3.hip): D's 144 -> 288 @127x127 and its data gradient
    const char* sel = FD_TUNE_GETENV("FDGAN_DEBUG_WD");
    char v = sel ? sel[0] : 'x';
    if (v != '0' && !narrow && a.Cin >= 32) {
      if (v == 'x') {
        const int waste128 = (cout_total + 127) / 128 * 128 - cout_total, waste144 = (cout_total + 143) / 144 * 144 - cout_total;
        v = cout_total <= 64 ? 'G' : (waste144 < waste128 ? 'H' : 'A');
      }
      if (a.grad_io) {
        if (v == 'G') FD_CONV_DISPATCH_W(4, 1, 0, 8, 2, 2, 2, 16, 1, 1, "conv4x4_wd64_bwd");
        if (v == 'H') FD_CONV_DISPATCH_W(4, 1, 0, 8, 3, 1, 3, 16, 1, 1, "conv4x4_wd144_bwd");
        FD_CONV_DISPATCH_W(4, 1, 0, 8, 2, 1, 4, 16, 1, 1, "conv4x4_wd128_bwd");
      }
      if (v == 'G') FD_CONV_DISPATCH_W(4, 1, 0, 8, 2, 2, 2, 16, 0, 1, "conv4x4_wd64");
      if (v == 'H') FD_CONV_DISPATCH_W(4, 1, 0, 8, 3, 1, 3, 16, 0, 1, "conv4x4_wd144");
      FD_CONV_DISPATCH_W(4, 1, 0, 8, 2, 1, 4, 16, 0, 1, "conv4x4_wd128");
    }
  }
  if (stride == 1 && a.grad_io) {   // backward data with the masked epilogue (fdgan_conv2d_bwd_data)
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
