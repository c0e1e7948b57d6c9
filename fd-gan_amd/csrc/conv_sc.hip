// conv_sc.hip -- "small Cin" forward convolutions: the layers that read an IMAGE.
//
//   3 -> 64, 3x3, stride 1, pad 1   FDGAN.conv_refin1 (/root/reference/models/dehaze1113.py:744, used at :760) and Vgg16.conv1_1
//                                   (/root/reference/myutils/vgg16.py:9, :28): three launches per training step
//   9 -> 36, 4x4, stride 2, pad 1   the Fusion-discriminator's first layer (/root/reference/models/dehaze1113.py:196): three launches
//
// On the generic implicit-GEMM kernel a 3-channel input is one 32-channel chunk with 29 dead channels per tap: nine k-steps of 32 for
// 27 real products (conv3x3_bn64: 107 us for a launch whose 151 MB of traffic take 27 us), and the 9-channel 4x4 is sixteen.  Here the
// k axis of the GEMM is (tap, channel) with the channels padded only to CP = 4 / 16:
//     K = 9 x 4 = 36 -> two k-steps,      K = 16 x 16 = 256 -> eight k-steps (9 of 16 channels real),
// and a B fragment (8 consecutive k of one output pixel) is two 8-byte pieces (two taps x four channels) or one 16-byte piece (half a
// tap) of the raw input tile in LDS, [pixel][CP] -- read with per-lane tap offsets, no im2col copy.  The A fragments for that k order are
// 16-byte (or two 8-byte) pieces of the library's chunk32 filter image as it is: global -> registers, no re-packing.  Workgroup = 8 x 32 output pixels x all output channels
// (CT tiles of 16), four waves of two output rows each; epilogue as conv_igemm's: bias, max(v, slope v), 16-byte row stores through a
// wave-private LDS transposition (or 8-byte stores for a ragged channel count), one row of batch-statistics partials.
// These launches are bound by their OUTPUT (134 MB / 21 MB); the matrix work is 4 us.
#include "conv_igemm.h"

namespace {

constexpr int SC_TH = 8, SC_TW = 32;

template <int KS, int ST, int CP, int CT>
struct ScCfg {
  static constexpr int KK = KS * KS, K = KK * CP, NS = (K + 31) / 32;          // k-steps
  static constexpr int IH = (SC_TH - 1) * ST + KS, IW = (SC_TW - 1) * ST + KS, NPIX = IH * IW;
  static constexpr int PB = CP * 2;                                             // bytes per staged pixel
  static constexpr int IN_B = (NPIX * PB + 15) / 16 * 16;
  static constexpr int W_B = 0;                                                 // (the filter goes global -> registers)
  static constexpr int UPP = PB >= 16 ? PB / 16 : 1;                            // 16-byte source pieces per pixel
  static constexpr int UNITS = NPIX * UPP, UPT = (UNITS + 255) / 256;
  static constexpr int PARTS = CP == 4 ? 2 : 1;                                 // LDS reads per B fragment
  static constexpr unsigned lds_bytes() { return IN_B + W_B + 4 * RowStore<CT>::BYTES + 4 * CT * 16 * 2 * 4; }
};

template <int KS, int ST, int CP, int CT>
__global__ __launch_bounds__(256, 2) void conv_sc_kernel(ConvArgs a) {
  using C = ScCfg<KS, ST, CP, CT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* in_lds = smem;
  char* w_lds = smem + C::IN_B;
  char* tb_all = w_lds + C::W_B;
  float* red = reinterpret_cast<float*>(tb_all + 4 * RowStore<CT>::BYTES);      // [4 waves][CT * 16][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  int tile = blockIdx.x;
  const int tx = tile % a.tiles_x;
  tile /= a.tiles_x;
  const int ty = tile % a.tiles_y, n = tile / a.tiles_y;
  const int oy0 = ty * SC_TH, ox0 = tx * SC_TW;
  // ---- the input tile: raw pixels (no prologue on this path), zeros outside the image
  const unsigned short* xn = a.x + (long long)n * a.x_sn;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 rin[C::UPT];
#pragma unroll
  for (int i = 0; i < C::UPT; ++i) {
    const int u = tid + i * 256, p = u / C::UPP, piece = u - p * C::UPP;
    const int py = p / C::IW, px = p - py * C::IW;
    const int gy = oy0 * ST - a.pad + py, gx = ox0 * ST - a.pad + px;
    const bool ok = u < C::UNITS && gy >= 0 && gy < a.Hs && gx >= 0 && gx < a.Ws;
    const u32x4 v = *reinterpret_cast<const u32x4*>(xn + (ok ? (long long)gy * a.x_sh + (long long)gx * a.x_sw + piece * 8 : 0));
    rin[i] = ok ? v : zero4;
  }
  // ---- the filter: A fragment (k-step s, cout tile ct), lane (i = cout & 15, kg): k = 32 s + 8 kg + e = tap * CP + c.  In the library's
  // chunk32 image (one chunk: [tap][tile16][lane' = (c >> 3) * 16 + (cout & 15)][c & 7]) the lane's 8 values are ONE 16-byte piece
  // (CP = 16: half a tap) or two 8-byte pieces of two taps' fragments (CP = 4): loaded straight into registers, all of them up front
  // (NS * CT fragments; round 5's first version re-ordered the image into LDS with 2-byte loads: 48 dependent rounds, 30 us).
  u32x4 wf[C::NS][CT];
#pragma unroll
  for (int s = 0; s < C::NS; ++s)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const bool tile_ok = c < a.ntile_total;
      if constexpr (CP == 4) {
        const int t0 = s * 8 + g * 2, t1 = t0 + 1;
        const u32x2 lo = *reinterpret_cast<const u32x2*>(a.w + ((long long)(t0 < C::KK ? t0 : 0) * a.ntile_total + (tile_ok ? c : 0)) * 512 + m * 8);
        const u32x2 hi = *reinterpret_cast<const u32x2*>(a.w + ((long long)(t1 < C::KK ? t1 : 0) * a.ntile_total + (tile_ok ? c : 0)) * 512 + m * 8);
        const bool ok0 = tile_ok && t0 < C::KK, ok1 = tile_ok && t1 < C::KK;
        wf[s][c] = u32x4{ok0 ? lo[0] : 0u, ok0 ? lo[1] : 0u, ok1 ? hi[0] : 0u, ok1 ? hi[1] : 0u};
      } else {
        const int k0 = s * 32 + g * 8, tap = k0 / CP, c0 = k0 - tap * CP;
        const bool ok = tile_ok && tap < C::KK;
        const u32x4 v = *reinterpret_cast<const u32x4*>(a.w + ((long long)(ok ? tap : 0) * a.ntile_total + (tile_ok ? c : 0)) * 512 + ((c0 >> 3) * 16 + m) * 8);
        wf[s][c] = ok ? v : zero4;
      }
    }
#pragma unroll
  for (int i = 0; i < C::UPT; ++i) {
    const int u = tid + i * 256, p = u / C::UPP, piece = u - p * C::UPP;
    if (u < C::UNITS) {
      if constexpr (CP == 4) *reinterpret_cast<u32x2*>(in_lds + p * 8) = u32x2{rin[i][0], rin[i][1]};
      else lds_write16(in_lds + p * C::PB + piece * 16, rin[i]);
    }
  }
  __syncthreads();
  // ---- per-lane tap offsets of the B fragments: k-step s, part q covers k = 32 s + 8 g (+ 4 q when CP == 4)
  int toff[C::NS][C::PARTS];
#pragma unroll
  for (int s = 0; s < C::NS; ++s)
#pragma unroll
    for (int q = 0; q < C::PARTS; ++q) {
      const int k0 = s * 32 + g * 8 + q * 4;
      int tap = k0 / CP;
      const int c0 = k0 - tap * CP;
      if (tap >= C::KK) tap = 0;      // dead k (their filter entries are zero): read a valid, finite pixel
      toff[s][q] = ((tap / KS) * C::IW + (tap % KS)) * C::PB + c0 * 2;
    }
  f32x4 acc[4][CT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // wave w: output rows 2 w, 2 w + 1; pixel tile t = (row r = t >> 1, column half t & 1)
  int pbase[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) pbase[t] = (((wave * 2 + (t >> 1)) * ST) * C::IW + ((t & 1) * 16 + m) * ST) * C::PB;
#pragma unroll
  for (int s = 0; s < C::NS; ++s) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      u32x4 xf;
      if constexpr (CP == 4) {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(in_lds + pbase[t] + toff[s][0]);
        const u32x2 hi = *reinterpret_cast<const u32x2*>(in_lds + pbase[t] + toff[s][1]);
        xf = u32x4{lo[0], lo[1], hi[0], hi[1]};
      } else {
        xf = lds_read16(in_lds + pbase[t] + toff[s][0]);
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[t][c] = fd_mfma<FmtA>(wf[s][c], xf, acc[t][c]);
    }
  }
  // ---- epilogue: bias, activation, stores, statistics (conv_igemm.h's forward epilogue on this tile shape)
  char* tb = tb_all + wave * RowStore<CT>::BYTES;
  const bool rowstore = a.y_vec16 && CT * 16 <= a.Cout;   // uniform
  float bv[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = c * 16 + g * 4 + r;
      const bool bok = a.bias != nullptr && co < a.CoutW;
      const float bval = (a.bias != nullptr ? a.bias : reinterpret_cast<const float*>(a.w))[bok ? co : 0];
      bv[c][r] = bok ? bval : 0.f;
    }
  float s1[CT][4], s2[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[c][r] = s2[c][r] = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = oy0 + wave * 2 + (t >> 1), colb = ox0 + (t & 1) * 16;
    float v[CT][4];
    const bool valid = row < a.Ho && colb + m < a.Wo;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float tt = acc[t][c][r] + bv[c][r];
        v[c][r] = fmaxf(tt, a.e_slope * tt);
        s1[c][r] += valid ? v[c][r] : 0.f;
        s2[c][r] += valid ? v[c][r] * v[c][r] : 0.f;
      }
    if (rowstore) {
      fd_store_row16<CT, false, FmtA>(a, tb, v, lane, 0, [&](int q) -> long long {
        return (row < a.Ho && colb + q < a.Wo) ? (long long)n * a.y_sn + (long long)row * a.y_sh + (long long)(colb + q) * a.y_sw : -1;
      });
    } else if (valid) {
      const long long off = (long long)n * a.y_sn + (long long)row * a.y_sh + (long long)(colb + m) * a.y_sw;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int cout0 = c * 16 + g * 4;
        if (cout0 < a.Cout) fd_store4<false, FmtA>(a, off, cout0, v[c]);
      }
    }
  }
  if (a.stats != nullptr) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = fd_row_sum16(s1[c][r]), t2 = fd_row_sum16(s2[c][r]);
        if (m == 0) {
          red[((wave * CT * 16) + c * 16 + g * 4 + r) * 2] = t1;
          red[((wave * CT * 16) + c * 16 + g * 4 + r) * 2 + 1] = t2;
        }
      }
    __syncthreads();
    for (int cl = tid; cl < CT * 16; cl += 256) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < 4; ++w_) {
        t1 += red[((w_ * CT * 16) + cl) * 2];
        t2 += red[((w_ * CT * 16) + cl) * 2 + 1];
      }
      float* dst = a.stats + ((long long)blockIdx.x * a.stats_cpad + cl) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

template <int KS, int ST, int CP, int CT>
int sc_launch(ConvArgs& a, long long nimg, FdConvInfo* info, long long stats_cap, bool dry, hipStream_t stream, const char* name) {
  using C = ScCfg<KS, ST, CP, CT>;
  a.tiles_x = (a.Wo + SC_TW - 1) / SC_TW;
  a.tiles_y = (a.Ho + SC_TH - 1) / SC_TH;
  dim3 grid((unsigned)(nimg * a.tiles_x * a.tiles_y), 1, 1);
  a.stats_cpad = CT * 16;
  const unsigned lds = C::lds_bytes();
  if (info) {
    info->stats_rows = grid.x;
    info->stats_cpad = a.stats_cpad;
    info->grid_x = grid.x;
    info->grid_y = 1;
    info->lds_bytes = lds;
  }
  if (dry) return FD_OK;
  if (stats_cap >= 0 && (long long)grid.x * a.stats_cpad * 2 > stats_cap)
    FD_FAIL(FD_EINVAL, "stats workspace too small: need %lld floats, have %lld", (long long)grid.x * a.stats_cpad * 2, stats_cap);
  auto kfn = &conv_sc_kernel<KS, ST, CP, CT>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(%s): %s", name, hipGetErrorString(e));
    attr_done = true;
  }
  return fd_launch(kfn, name, grid, dim3(256), lds, a, stream);
}

}  // namespace

// which of the two shapes (0: none): raw fp16 input (no prologue, no pooling), NHWC 16-bit output without upsampling, the chunk32 filter
// image of a single chunk
int conv_sc_variant(const ConvArgs& a, int cout_total, int ksize, int stride, bool pool) {
  if (pool || a.grad_io || a.pro_mode != 0 || a.out_nchw_f32 || a.upsample || FD_TUNE_GETENV("FDGAN_DEBUG_NO_SC") != nullptr) return 0;
  if (a.nchunk != 1) return 0;
  if (ksize == 3 && stride == 1 && a.pad == 1 && a.Cin <= 4 && cout_total > 32 && cout_total <= 64 && a.x_sw >= 8) return 1;
  if (ksize == 4 && stride == 2 && a.pad == 1 && a.Cin <= 16 && cout_total > 16 && cout_total <= 48 && a.x_sw >= 16) return 2;
  return 0;
}

int conv_sc_launch(int variant, ConvArgs& a, long long nimg, FdConvInfo* info, long long stats_cap, bool dry, hipStream_t stream) {
  if (variant == 1) return sc_launch<3, 1, 4, 4>(a, nimg, info, stats_cap, dry, stream, "conv3x3_sc4_bn64");
  return sc_launch<4, 2, 16, 3>(a, nimg, info, stats_cap, dry, stream, "conv4x4s2_sc16_bn48");
}
