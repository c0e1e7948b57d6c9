// conv1x1_xs.hip -- 1x1 convolution, "x-stream" form (the dense-layer bottlenecks, the
// transitions with their pooled BN+ReLU prologue, the decoder 1x1s).
//
// A 1x1 conv is a plain GEMM  D[cout][pixel] = W[cout][cin] * A[cin][pixel]  whose pixel
// operand is used by exactly ONE wave (waves split the pixel dimension, every wave covers
// all BN output channels).  So the activations never need LDS:
//   * each lane loads its MFMA B-fragment straight from HBM: lane (m = l&15, g = l>>4) reads
//     32 contiguous bytes (channels g*16 .. g*16+15) of pixel m per 64-channel k-step, so
//     4 lanes consume one full 128-byte line per pixel; the BatchNorm affine + ReLU (and
//     the 2x2 average of the transition) run on those registers;
//   * the k-index mapping inside a fragment is free as long as the filter uses the same
//     one, so the filter is packed for exactly this mapping ("x64" layout) and is loaded
//     ONCE per workgroup into LDS, where all four waves read it (lane-linear, 16 B/lane);
//   * workgroups are persistent: they walk pixel tiles with a grid stride, keep the
//     filter and the BN scale/shift resident, prefetch the next k-step's fragments while
//     the MFMAs of the current one run AND the next tile's first k-step while the epilogue
//     of the current tile runs, and emit ONE row of batch-statistics partials per
//     workgroup (<= 512 rows for fdgan_bn_finalize instead of one per tile).
// No barrier inside the tile loop.
#include <stdlib.h>

#include "conv_igemm.h"

template <int POOL, int PT, int CT, int NW_>
struct XsCfg {
  static constexpr int NW = NW_, NT = 64 * NW_;
  static constexpr int SC = CT > 4 ? 4 : CT;   // channel tiles per row-store pass (<= one 128-byte line)
  static constexpr int WPX = PT * 16;          // pixels per wave per tile
  static constexpr int TILE_PX = NW * WPX;
  static constexpr int BN = CT * 16;
  static constexpr int NL = POOL ? 4 : 1;      // source pixels per output pixel
  // LDS: [kgroup k-steps of the filter][scale, shift for all nks*64 channels][stats][row-store staging]
  __host__ __device__ static unsigned w_bytes(int kgroup) { return (unsigned)kgroup * 2 * CT * 1024; }
  __host__ __device__ static unsigned lds_bytes(int nks, int kgroup) {
    return w_bytes(kgroup) + nks * 64 * 8 + NW * BN * 2 * 4 + NW * RowStore<SC>::BYTES;
  }
  // largest k-group (<= nks) whose filter slice fits next to the fixed parts
  static int max_kgroup(int nks) {
    const long long fixed = (long long)nks * 64 * 8 + NW * BN * 2 * 4 + NW * RowStore<SC>::BYTES;
    long long g = (160ll * 1024 - fixed) / (2 * CT * 1024);
    return (int)(g < nks ? g : nks);
  }
};

template <int POOL, int PT, int CT, int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv1x1_xs_kernel(ConvArgs a) {
  using C = XsCfg<POOL, PT, CT, NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w_lds = smem;                                                   // [nks][2][CT][1 KiB]
  float* sc_lds = reinterpret_cast<float*>(smem + C::w_bytes(a.kgroup)); // [nks*64]
  float* sh_lds = sc_lds + a.nks * 64;
  float* red = sh_lds + a.nks * 64;                                     // [NW][BN][2]
  char* tb = reinterpret_cast<char*>(red + C::NW * C::BN * 2) + (threadIdx.x >> 6) * RowStore<C::SC>::BYTES;  // row-store staging

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int by = blockIdx.y;

  // ---- once per workgroup: BN fold, filter -> LDS, zero the statistics accumulators
  fd_fold_bn(a, sc_lds, sh_lds, a.nks * 64, tid, C::NT);
  // filter k-steps [k0, k0+cnt) -> LDS (lane-linear fragments); all threads
  auto load_w = [&](int k0, int cnt) {
    const int nunits = cnt * 2 * CT * 64;
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    for (int u = tid; u < nunits; u += C::NT) {
      const int kj = u / (CT * 64), rem = u - kj * (CT * 64);
      const int tile16 = by * CT + (rem >> 6);
      const bool ok = tile16 < a.ntile_total;
      const u32x4 v = *reinterpret_cast<const u32x4*>(
          a.w + (ok ? ((long long)(k0 * 2 + kj) * a.ntile_total + tile16) * 512 + (rem & 63) * 8 : 0));
      lds_write16(w_lds + u * 16, ok ? v : z4);
    }
  };
  const bool resident = a.kgroup >= a.nks;   // the whole filter stays in LDS for every tile
  if (resident) load_w(0, a.nks);
  for (int i = tid; i < C::NW * C::BN * 2; i += C::NT) red[i] = 0.f;
  __syncthreads();

  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const char* wfrag = w_lds + lane * 16;
  const int cmax = a.Cin8 * 8;
  const unsigned HW = (unsigned)a.Ho * (unsigned)a.Wo;
  const int up = a.upsample ? 2 : 1;
  const bool plain_epi = a.bias == nullptr && a.e_slope == 1.f;

  // pixel -> element offsets.  A pixel past the end is clamped to pixel 0 (a valid address).
  auto x_offset = [&](unsigned px) -> unsigned {
    const unsigned q = px < a.P ? px : 0u;
    if (a.x_dense) return q * (unsigned)a.x_sw + g * 16;
    const unsigned n = q / HW, r = q - n * HW;
    const unsigned oy = r / (unsigned)a.Wo, ox = r - oy * (unsigned)a.Wo;
    return (unsigned)(n * (unsigned long long)a.x_sn) + (POOL ? 2 * oy : oy) * (unsigned)a.x_sh +
           (POOL ? 2 * ox : ox) * (unsigned)a.x_sw + g * 16;
  };
  auto y_offset = [&](unsigned px) -> unsigned {
    if (a.y_dense) return px * (unsigned)a.y_sw;
    const unsigned n = px / HW, r = px - n * HW;
    const unsigned oy = r / (unsigned)a.Wo, ox = r - oy * (unsigned)a.Wo;
    return (unsigned)(n * (unsigned long long)a.y_sn) + (up * oy) * (unsigned)a.y_sh + (up * ox) * (unsigned)a.y_sw;
  };

  u32x4 raw[PT][2][C::NL];
  unsigned xoff[PT];   // element offsets (< 2^32, checked on the host)
  auto load = [&](int ks, int j) {   // half a k-step: this lane's j-th 16 bytes of every pixel
    const int cb = ks * 64 + g * 16 + j * 8;
    const int coff = cb < cmax ? ks * 64 + j * 8 : -g * 16;   // past Cin: re-read channel 0 (masked later)
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const unsigned short* src = a.x + xoff[p] + coff;
      raw[p][j][0] = *reinterpret_cast<const u32x4*>(src);
      if (POOL) {
        raw[p][j][1] = *reinterpret_cast<const u32x4*>(src + a.x_sw);
        raw[p][j][2] = *reinterpret_cast<const u32x4*>(src + a.x_sh);
        raw[p][j][3] = *reinterpret_cast<const u32x4*>(src + a.x_sh + a.x_sw);
      }
    }
  };

  unsigned tile = blockIdx.x;
  if (tile < (unsigned)a.ntiles) {
#pragma unroll
    for (int p = 0; p < PT; ++p) xoff[p] = x_offset(tile * C::TILE_PX + wave * C::WPX + p * 16 + m);
    load(0, 0);
    load(0, 1);
  }
  for (; tile < (unsigned)a.ntiles; tile += gridDim.x) {
    const unsigned px0 = tile * C::TILE_PX + wave * C::WPX + m;   // pixel of p = 0
    const bool full = tile * C::TILE_PX + C::TILE_PX <= a.P;       // uniform: no ragged pixels in this tile
    const unsigned tnext = tile + gridDim.x;
    const bool has_next = tnext < (unsigned)a.ntiles;

    f32x4 acc[PT][CT];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int ks = 0; ks < a.nks; ++ks) {
      const bool last = ks + 1 == a.nks;
      const int kl = resident ? ks : ks % a.kgroup;   // k-step index inside the LDS-resident slice
      if (!resident && kl == 0) {                     // stream the next filter slice (uniform per workgroup)
        __syncthreads();                              // every wave is done with the previous slice
        const int left = a.nks - ks;
        load_w(ks, left < a.kgroup ? left : a.kgroup);
        __syncthreads();
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // registers -> activated fp16 fragments; the freed registers take the next loads (next k-step,
        // or the next tile's first k-step), which stay in flight while the MFMAs / the epilogue run
        f16x8 xf[PT];
        const int cb = ks * 64 + g * 16 + j * 8;
        const bool cok = cb < cmax;
        const float* sc = sc_lds + cb;
        const float* sh = sh_lds + cb;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          u32x4 v;
          if (POOL) {
            f32x8 f = fd_affine_act(raw[p][j][0], sc, sh, a.p_slope);
            f += fd_affine_act(raw[p][j][1], sc, sh, a.p_slope);
            f += fd_affine_act(raw[p][j][2], sc, sh, a.p_slope);
            f += fd_affine_act(raw[p][j][3], sc, sh, a.p_slope);
            v = fd_pack8(f * 0.25f);
          } else if (a.pro_mode == 0) {   // uniform: no transform at all
            v = raw[p][j][0];
          } else {
            v = fd_xform8(raw[p][j][0], sc, sh, a.p_slope);
          }
          // a pixel past the end / a channel group past Cin must contribute exactly zero
          const bool ok = cok && (full || px0 + p * 16 < a.P);
          xf[p] = __builtin_bit_cast(f16x8, ok ? v : zero4);
        }
        // The next fragments are requested UNCONDITIONALLY (round 5): inside `if (!last) ... else if (has_next) ...` the loads sat in
        // conditional blocks, hipcc drained them at the end of each block (s_waitcnt vmcnt(0) right behind the loads: tools/
        // loop_wait_audit.py) and this "prefetch" overlapped nothing.  The last k-step of the last tile re-reads its own first
        // k-step (a valid address, never used).
        if (last && j == 0) {   // re-target the source offsets (no loads in this block); this tile issues no further loads
          const unsigned tn = has_next ? tnext : tile;
#pragma unroll
          for (int p = 0; p < PT; ++p) xoff[p] = x_offset(tn * C::TILE_PX + wave * C::WPX + p * 16 + m);
        }
        load(last ? 0 : ks + 1, j);
        constexpr int CH = CT > 4 ? 4 : CT;  // filter fragments live at a time
#pragma unroll
        for (int c0 = 0; c0 < CT; c0 += CH) {
          f16x8 wf[CH];
#pragma unroll
          for (int c = 0; c < CH; ++c)
            wf[c] = __builtin_bit_cast(f16x8, lds_read16(wfrag + ((kl * 2 + j) * CT + c0 + c) * 1024));
#pragma unroll
          for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int c = 0; c < CH; ++c)
              acc[p][c0 + c] = fd_mfma_a(wf[c], xf[p], acc[p][c0 + c]);
        }
      }
    }

    // ---- epilogue: bias, activation, row stores, statistics
    const int cbase = by * C::BN;
    const bool rowstore = a.y_vec16 && a.y_dense && cbase + C::BN <= a.Cout;   // uniform
    if (plain_epi && rowstore) {
      // fast path: no bias, identity activation, whole rows through the LDS staging area
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const unsigned pxt = tile * C::TILE_PX + wave * C::WPX + p * 16;
#pragma unroll
        for (int c0 = 0; c0 < CT; c0 += C::SC) {
          float v[C::SC][4];
#pragma unroll
          for (int c = 0; c < C::SC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[c][r] = acc[p][c0 + c][r];
          fd_store_row16<C::SC>(a, tb, v, lane, cbase + c0 * 16, [&](int q) -> long long {
            return pxt + q < a.P ? (long long)(pxt + q) * a.y_sw : -1;
          });
        }
      }
      if (a.stats != nullptr) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int p = 0; p < PT; ++p) {
            const bool pok = full || px0 + p * 16 < a.P;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = pok ? acc[p][c][r] : 0.f;
              s1[r] += v;
              s2[r] = fmaf(v, v, s2[r]);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[r] = fd_row_sum16(s1[r]);
            s2[r] = fd_row_sum16(s2[r]);
          }
          if (m == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* d = red + ((wave * C::BN) + c * 16 + g * 4 + r) * 2;
              d[0] += s1[r];
              d[1] += s2[r];
            }
          }
        }
      }
    } else {
      unsigned yoff[PT];
#pragma unroll
      for (int p = 0; p < PT; ++p) yoff[p] = y_offset(px0 + p * 16 < a.P ? px0 + p * 16 : 0u);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int cout0 = cbase + c * 16 + g * 4;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        float bv[4];
#pragma unroll
        // (conditional on purpose: the unconditional clamped form of conv_igemm.h costs this kernel 14 registers at the end of its
        // pixel loop -- 42 -> 290 scratch instructions over the instantiations, the pooled transitions 0 -> 64; tools/scratch_audit.py)
        for (int r = 0; r < 4; ++r) bv[r] = (a.bias != nullptr && cout0 + r < a.CoutW) ? a.bias[cout0 + r] : 0.f;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          const bool pok = px0 + p * 16 < a.P;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = acc[p][c][r] + bv[r];
            v[r] = fmaxf(t, a.e_slope * t);
            s1[r] += pok ? v[r] : 0.f;
            s2[r] += pok ? v[r] * v[r] : 0.f;
          }
          if (pok && cout0 < a.Cout) fd_store4(a, yoff[p], cout0, v);
        }
        if (a.stats != nullptr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[r] = fd_row_sum16(s1[r]);
            s2[r] = fd_row_sum16(s2[r]);
          }
          if (m == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* d = red + ((wave * C::BN) + c * 16 + g * 4 + r) * 2;
              d[0] += s1[r];
              d[1] += s2[r];
            }
          }
        }
      }
    }
  }

  if (a.stats != nullptr) {
    __syncthreads();
    for (int cl = tid; cl < C::BN; cl += C::NT) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < C::NW; ++w_) {
        t1 += red[(w_ * C::BN + cl) * 2];
        t2 += red[(w_ * C::BN + cl) * 2 + 1];
      }
      float* dst = a.stats + ((long long)blockIdx.x * a.stats_cpad + by * C::BN + cl) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

// ---------------------------------------------------------------------------------
static int g_num_cus = 0;
static int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_num_cus = prop.multiProcessorCount;
    if (g_num_cus <= 0) g_num_cus = 256;
  }
  return g_num_cus;
}

#define FD_XS_LAUNCH(POOL_, PT_, CT_, NW_, PERCU_, NAME_)                                                                   \
  do {                                                                                                          \
    using C = XsCfg<POOL_, PT_, CT_, NW_>;                                                                       \
    a.kgroup = C::max_kgroup(a.nks);                                                                            \
    if (a.kgroup < 1) FD_FAIL(FD_EUNSUPPORTED, "conv1x1_xs: Cin too large for LDS");                           \
    const unsigned lds = C::lds_bytes(a.nks, a.kgroup);                                                         \
    a.ntiles = (int)((a.P + C::TILE_PX - 1) / C::TILE_PX);                                                      \
    const int per_cu = PERCU_;                                                                                  \
    const int ncu = fd_cus(dry ? 256 : num_cus());                                                                     \
    const unsigned gy = (unsigned)((cout_total + C::BN - 1) / C::BN);                                           \
    long long gx = (long long)per_cu * ncu / gy;                                                                \
    if (gx < 1) gx = 1;                                                                                         \
    if (gx > a.ntiles) gx = a.ntiles;                                                                           \
    dim3 grid((unsigned)gx, gy, 1), block(C::NT, 1, 1);                                                             \
    a.stats_cpad = gy * C::BN;                                                                                  \
    if (info) {                                                                                                 \
      info->stats_rows = grid.x;                                                                                \
      info->stats_cpad = a.stats_cpad;                                                                          \
      info->grid_x = grid.x;                                                                                    \
      info->grid_y = grid.y;                                                                                    \
      info->lds_bytes = lds;                                                                                    \
    }                                                                                                           \
    if (dry) return FD_OK;                                                                                      \
    auto kfn = &conv1x1_xs_kernel<POOL_, PT_, CT_, NW_>;                                                        \
    static bool attr_done = false;                                                                              \
    if (!attr_done) {                                                                                           \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
      if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(%s): %s", NAME_, hipGetErrorString(e));     \
      attr_done = true;                                                                                         \
    }                                                                                                           \
    if (stats_cap >= 0 && (long long)grid.x * a.stats_cpad * 2 > stats_cap)                                     \
      FD_FAIL(FD_EINVAL, "stats workspace too small: need %lld floats, have %lld",                              \
              (long long)grid.x * a.stats_cpad * 2, stats_cap);                                                 \
    return fd_launch(kfn, NAME_, grid, block, lds, a, stream);                                                  \
  } while (0)

// 4 waves per workgroup (two workgroups per CU) while the LDS-resident filter leaves room for
// two; otherwise ONE 8-wave workgroup per CU (still two waves per SIMD), with 32-pixel wave
// tiles when there are too few pixels to give every CU a 512-pixel tile.  Filters larger than
// LDS are streamed in k-groups (two barriers per group).
bool conv1x1_xs_fits(int cout_total, int cin) {
  (void)cout_total;
  return cin >= 1 && cin <= 4096;   // scale/shift for every input channel must fit LDS
}

int conv_dispatch_k1_xs(ConvArgs& a, long long nimg, int cout_total, bool pool, FdConvInfo* info,
                        long long stats_cap, bool dry, hipStream_t stream) {
  a.nks = (a.Cin + 63) / 64;
  const long long P64 = nimg * (long long)a.Ho * a.Wo;
  if (P64 >= (1ll << 31)) FD_FAIL(FD_EUNSUPPORTED, "conv1x1_xs: more than 2^31 output pixels");
  if (nimg * a.x_sn >= (1ll << 32) || nimg * a.y_sn >= (1ll << 32))
    FD_FAIL(FD_EUNSUPPORTED, "conv1x1_xs: tensors beyond 2^32 elements");
  a.P = (unsigned)P64;
  a.x_dense = !pool && a.x_sh == (long long)a.Ws * a.x_sw && a.x_sn == (long long)a.Hs * a.x_sh;
  a.y_dense = !a.upsample && !a.out_nchw_f32 && a.y_sh == (long long)a.Wo * a.y_sw &&
              a.y_sn == (long long)a.Ho * a.y_sh;
  if (!conv1x1_xs_fits(cout_total, a.Cin)) FD_FAIL(FD_EUNSUPPORTED, "conv1x1_xs: filter does not fit LDS");
  // the bottleneck shape (-> 128 channels, dense tensors, plain epilogue) streams through LDS-DMA
  if (conv1x1_ds_fits(a, cout_total, pool, FD_WLAYOUT_X64) && FD_TUNE_GETENV("FDGAN_DEBUG_NO_DS") == nullptr)
    return conv_dispatch_k1_ds(a, info, stats_cap, dry, stream);
  const int nks = a.nks;
  const long long big_tiles = (a.P + 511) / 512;
  const int ncu_ = fd_cus(dry ? 256 : num_cus());
  if (pool) {
    if (cout_total <= 32) FD_XS_LAUNCH(1, 2, 2, 4, 2, "conv1x1_xs_pool_bn32");
    if (cout_total <= 64) FD_XS_LAUNCH(1, 2, 4, 4, 2, "conv1x1_xs_pool_bn64");
    if (XsCfg<1, 2, 8, 4>::lds_bytes(nks, nks) <= 80 * 1024) FD_XS_LAUNCH(1, 2, 8, 4, 2, "conv1x1_xs_pool_bn128");
    FD_XS_LAUNCH(1, 2, 8, 8, 1, "conv1x1_xs_pool_bn128_w8");
  }
  if (cout_total <= 32) FD_XS_LAUNCH(0, 4, 2, 4, 2, "conv1x1_xs_bn32");
  if (cout_total <= 64) FD_XS_LAUNCH(0, 4, 4, 4, 2, "conv1x1_xs_bn64");
  if (XsCfg<0, 4, 8, 4>::lds_bytes(nks, nks) <= 80 * 1024) FD_XS_LAUNCH(0, 4, 8, 4, 2, "conv1x1_xs_bn128");
  if (big_tiles >= ncu_) FD_XS_LAUNCH(0, 4, 8, 8, 1, "conv1x1_xs_bn128_w8");
  FD_XS_LAUNCH(0, 2, 8, 8, 1, "conv1x1_xs_bn128_w8p2");
}
