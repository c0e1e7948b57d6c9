// 3x3 stride-1 convolutions (dense-layer growth convs, refine convs, D, VGG16).
#include <stdlib.h>

#include "conv_igemm.h"

int conv_dispatch_k3(ConvArgs& a, long long nimg, int cout_total, int stride, bool pool, FdConvInfo* info,
                     long long stats_cap, bool dry, hipStream_t stream) {
  const bool narrow = cout_total <= 32;
  if (pool) FD_FAIL(FD_EUNSUPPORTED, "pool2 prologue needs a 1x1 stride-1 conv");
  if (stride != 1) FD_FAIL(FD_EUNSUPPORTED, "3x3 conv with stride %d", stride);
  if (!a.grad_io && conv3x3_rs_fits(a, cout_total) && FD_TUNE_GETENV("FDGAN_DEBUG_NO_RS") == nullptr)
    return conv_dispatch_k3_rs(a, nimg, cout_total, info, stats_cap, dry, stream);
  if (!a.grad_io && narrow && a.pad == 1 && conv3x3_pw_fits(cout_total, a.Cin) && FD_TUNE_GETENV("FDGAN_DEBUG_NO_PW") == nullptr)
    return conv_dispatch_k3_pw(a, nimg, cout_total, info, stats_cap, dry, stream);
  // MFMA-bound shapes (VGG16, D, the refine convs, the dy blocks' 3x3) and their data gradients: filter-direct kernels
  // (conv_igemm.h, WD = 1), 8 x 16 output pixels per wave, 2-3 workgroups per CU:
  //   wdA: 128 output channels per workgroup (4 waves x 32), wdG: 64 (2 x 2 waves, 16 x 16 pixels),
  //   wdH: 144 (3 waves x 48): D's 72 -> 144 -> 288 widths without padding waste.
  // (Round 4, measured and not kept: 8 waves = 16 x 16 pixels x 128 channels per workgroup, i.e. half of wdA's filter bytes from L2 per
  // pixel -- 2048 workgroups each pull the whole 295 KB filter of a 128 -> 128 layer -- VGG16's forward 0.77 -> 0.86 ms: the filter
  // traffic is not what holds these kernels at 1.0-1.1 PFLOP/s.  Nor is occupancy: 4 rows per wave on 8 waves (110 registers, four
  // waves per SIMD instead of three) runs the same layers at 0.87-0.90 PFLOP/s, 0.89 ms; and 64 output channels per wave on 2 waves
  // (half the input-fragment reads per MFMA) needs 256 registers, spills 28-31 and runs at 0.68-0.95 PFLOP/s, 0.99 ms.)
  {
    const char* sel = FD_TUNE_GETENV("FDGAN_DEBUG_WD");   // tuning aid: 0 forces the LDS-staged-filter kernels, A/G/H a variant
    char v = sel ? sel[0] : 'x';
    if (v != '0' && !narrow && a.Cin >= 32) {
      if (v == 'x') {
        const int waste128 = (cout_total + 127) / 128 * 128 - cout_total, waste144 = (cout_total + 143) / 144 * 144 - cout_total;
        v = cout_total <= 64 ? 'G' : (waste144 < waste128 ? 'H' : 'A');
      }
      if (a.grad_io) {
        if (v == 'G') FD_CONV_DISPATCH_W(3, 1, 0, 8, 2, 2, 2, 9, 1, 1, "conv3x3_wd64_bwd");
        if (v == 'H') FD_CONV_DISPATCH_W(3, 1, 0, 8, 3, 1, 3, 9, 1, 1, "conv3x3_wd144_bwd");
        FD_CONV_DISPATCH_W(3, 1, 0, 8, 2, 1, 4, 9, 1, 1, "conv3x3_wd128_bwd");
      }
      if (v == 'G') FD_CONV_DISPATCH_W(3, 1, 0, 8, 2, 2, 2, 9, 0, 1, "conv3x3_wd64");
      if (v == 'H') FD_CONV_DISPATCH_W(3, 1, 0, 8, 3, 1, 3, 9, 0, 1, "conv3x3_wd144");
      FD_CONV_DISPATCH_W(3, 1, 0, 8, 2, 1, 4, 9, 0, 1, "conv3x3_wd128");
    }
  }
  const bool mid = cout_total <= 64;   // 64 output channels per workgroup: the 128-wide tile would idle half its MFMAs (VGG16 conv1_2: 197 us)
  if (a.grad_io) {   // backward data with the masked epilogue (fdgan_conv2d_bwd_data)
    if (narrow) FD_CONV_DISPATCH_X(3, 1, 0, 4, 2, 4, 1, 9, 1, "conv3x3_bn32_bwd");
    if (mid) FD_CONV_DISPATCH_X(3, 1, 0, 4, 4, 4, 1, 1, 1, "conv3x3_bn64_bwd");
    FD_CONV_DISPATCH_X(3, 1, 0, 4, 8, 4, 1, 1, 1, "conv3x3_bn128_bwd");
  }
  if (narrow) FD_CONV_DISPATCH(3, 1, 0, 4, 2, 4, 1, 9, "conv3x3_bn32");
  if (mid) FD_CONV_DISPATCH(3, 1, 0, 4, 4, 4, 1, 1, "conv3x3_bn64");
  FD_CONV_DISPATCH(3, 1, 0, 4, 8, 4, 1, 1, "conv3x3_bn128");
}
