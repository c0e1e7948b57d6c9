// conv1x1_ds.hip -- the dense-layer bottleneck: 1x1 convolution, Cin (64 .. 1024, a channel prefix of
// the block's concat buffer) -> 128 channels, BatchNorm + ReLU prologue, batch statistics of the
// result.  58 of netG's launches and half of its forward time.  "ds" = DMA-streamed.
//
// A 1x1 conv is a GEMM  D[cout][pixel] = W[cout][cin] * A[cin][pixel]  that reads every activation
// exactly once, so all that matters is how the activations stream in.  Its predecessor
// (conv1x1_xs) loads MFMA B fragments straight from HBM into registers: 16 pixels x 64 B per wave
// instruction, a few instructions in flight per wave -- 2.7 TB/s marginal at 64x64, 3.7 TB/s at 256x256.
// Here every workgroup owns 256 consecutive pixels per tile and walks the channel axis in stages of
// 64 channels; a stage (256 px x 128 B of activations + the 16 KiB filter slice) is fetched by
// LDS-DMA two stages ahead (tools/ubench/kstream.hip: this pattern streams at 5.8-6.9 TB/s).
//
//   * 8 waves, all alike.  Wave w owns pixels [32w, 32w+32) of the tile: it issues the DMA of exactly
//     those pixels (4 instructions per stage) plus 2 of the 16 filter fragments, and consumes them:
//     per stage 4 raw B fragments from LDS -> BatchNorm + ReLU in registers (20 VALU per fragment,
//     every element once) -> 32 MFMAs against the 16 A fragments of the slice.  Only the filter
//     is shared between waves: one barrier per stage.
//   * LDS image of a stage: [256 px][8 slots of 16 B], slot = chunk ^ ((px >> 1) & 5), applied on the
//     DMA source address.  The filter keeps the "x64" fragment order of conv1x1_xs (lane group g owns
//     channels 64 ks + 16 g + 8 j), so the B read of MFMA j takes chunk 2g + j; with this swizzle the
//     16 lanes of every ds_read_b128 lane group hit 16 distinct 16-byte bank groups.
//   * the epilogue (statistics in registers across all tiles of the workgroup, 256-byte row stores
//     staged through the wave's own, just-consumed activation region) overlaps the two stages
//     already in flight for the next tile.
// Reference call sites replaced: torchvision _DenseLayer conv1 (densenet.py) as used by
// /root/reference/models/dehaze1113.py:711-722, and BottleneckBlockdy.conv1 (:262).
#include <stdlib.h>

#include "conv_igemm.h"

namespace {

constexpr int DS_W_B = 16 * 1024;   // one stage of the filter: 2 k32-steps x 8 cout tiles
constexpr int DS_NBUF = 3;          // two stages in flight + the one being consumed
constexpr int DS_CT = 8;            // cout tiles

// NW waves, each owning PT pixel tiles of 16 pixels
template <int NW_, int PT_>
struct DsCfg {
  static constexpr int NW = NW_, PT = PT_, NT = 64 * NW_;
  static constexpr int WPX = PT_ * 16;              // pixels per wave
  static constexpr int PX = NW_ * WPX;              // pixels per tile
  static constexpr int ACT_B = PX * 128;            // one stage of activations: 64 channels per pixel
  static constexpr int STAGE_B = ACT_B + DS_W_B;
  static constexpr int AI = 2 * PT_;                // activation DMA instructions per wave per stage (1 KiB each)
  static constexpr int FI = NW_ >= 16 ? 1 : 2;      // filter fragments per wave per stage (16 in all; NW < 16: some twice)
  static constexpr int IPS = AI + FI;               // DMA instructions per wave per stage
  static constexpr int SC = PT_ >= 2 ? 4 : 2;       // cout tiles per row-store pass (staging must fit the wave's region)
  static constexpr int NSTORE = PT_ * (DS_CT / SC) * (16 / RowStore<SC>::PPI);
  static_assert(RowStore<SC>::BYTES <= WPX * 128, "row-store staging must fit the wave's activation region");
  __host__ __device__ static unsigned lds_bytes(int nks) { return DS_NBUF * STAGE_B + nks * 64 * 8 + NW * 128 * 2 * 4; }
};

__device__ __forceinline__ void ds_dma16(const unsigned short* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void ds_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// barrier that leaves LDS-DMA in flight (see conv3x3_rs.hip)
__device__ __forceinline__ void ds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Phase skips (tuning builds only, FDGAN_DEBUG_PHASES; results wrong): 1 no filter DMA after the pipeline fill, 2 no MFMAs, 4 no
// prologue transform, 8 no output stores, 16 no activation DMA after the pipeline fill, 32 no statistics.  (s_setprio(1) around
// the MFMA block, tried the same way in round 4: no effect, 127.7 vs 127.7 us at 256^2 c128.)
#ifdef FDGAN_TUNING
#define DS_SKIP(bit) ((a.dbg_skip & (bit)) != 0)
#else
#define DS_SKIP(bit) false
#endif

struct DsTimer {   // measurement aid (tools/conv_bench.py FDGAN_TIMING=1): s_memtime per phase, workgroup 0
  bool on;
  unsigned long long t[6], last;
  __device__ __forceinline__ void start(bool enable) {
    on = enable;
    for (int k = 0; k < 6; ++k) t[k] = 0;
    last = on ? __builtin_amdgcn_s_memtime() : 0;
  }
  __device__ __forceinline__ void stamp(int k) {
    if (on) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      t[k] += now - last;
      last = now;
    }
  }
};

// XMODE: 0 raw, 1 BatchNorm + ReLU, 2 affine + max(v, slope*v)
template <int XMODE, int NW, int PT>
__global__ __launch_bounds__(64 * NW) void conv1x1_ds_kernel(ConvArgs a) {
  using C = DsCfg<NW, PT>;
  constexpr int DS_NT = C::NT, DS_PX = C::PX, DS_ACT_B = C::ACT_B, DS_STAGE_B = C::STAGE_B, DS_PT = PT, DS_NW = NW;
  constexpr int DS_IPS = C::IPS, DS_NSTORE = C::NSTORE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* stage0 = smem;                                                         // [NBUF][act | filter]
  float* sc_lds = reinterpret_cast<float*>(smem + DS_NBUF * DS_STAGE_B);       // [nks*64]
  float* sh_lds = sc_lds + a.nks * 64;
  float* red = sh_lds + a.nks * 64;                                            // [NW][128][2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  DsTimer tm;
  tm.start(a.dbg != nullptr && blockIdx.x == 0);
  // (issuing the first two stages before this fold was measured 3-10 % SLOWER: hipcc then keeps the fold's
  // global loads behind the in-flight LDS-DMA)
  fd_fold_bn(a, sc_lds, sh_lds, a.nks * 64, tid, DS_NT);
  __syncthreads();
  tm.stamp(5);

  const int my_tiles = (int)blockIdx.x < a.ntiles ? (a.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int total = my_tiles * a.nks;
  const int cmax = a.Cin8 * 8;
  // (Tried in round 4: masked channel chunks reading their OWN address up to the end of the last 128-byte line instead of all
  // parking on chunk 0 -- a 96-channel prefix takes longer than a 128-channel one, 135.6 vs 124.5 us at 256^2 -- made no
  // difference (133.1 vs 136.2 us, inside the run-to-run drift), and neither did one shared address for all of them.  tools/ds_sweep.py has the per-phase table.)

  // ---- DMA maps.  Activation instruction i of this wave covers LDS positions [(4 wave + i) KiB, +1 KiB):
  // local pixel 32 wave + 8 i + lane / 8, slot lane % 8, holding channel chunk slot ^ ((px >> 1) & 5).
  int dpx[C::AI], dch[C::AI];
#pragma unroll
  for (int i = 0; i < C::AI; ++i) {
    dpx[i] = wave * C::WPX + i * 8 + (lane >> 3);
    dch[i] = ((lane & 7) ^ ((dpx[i] >> 1) & 5)) * 8;   // element offset of the chunk inside the 64-channel step
  }
  unsigned dsrc[C::AI];   // element offset of the lane's pixel in x for the tile being fetched
  auto retarget = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::AI; ++i) {
      const unsigned px = (unsigned)tile * DS_PX + (unsigned)dpx[i];
      dsrc[i] = (px < a.P ? px : 0u) * (unsigned)a.x_sw;   // past the end: pixel 0 (masked by the consumer)
    }
  };
  int f_tile = (int)blockIdx.x, f_ks = 0;   // stage to be fetched next
  int issued = 0;
  auto issue = [&](int buf) __attribute__((always_inline)) {
    char* dst = stage0 + buf * DS_STAGE_B;
    if (f_ks == 0) retarget(f_tile);
    const bool fill = issued < 2;
    ++issued;
    if (fill || !DS_SKIP(16))
#pragma unroll
    for (int i = 0; i < C::AI; ++i) {
      const int ch = f_ks * 64 + dch[i];
      ds_dma16(a.x + dsrc[i] + (ch < cmax ? ch : 0), dst + (wave * C::AI + i) * 1024);   // past Cin: chunk 0 (masked)
    }
    if (fill || !DS_SKIP(1))
#pragma unroll
    for (int f = 0; f < C::FI; ++f) {
      // fragment (j = fi >> 3, cout tile fi & 7) of k-step f_ks; with fewer than 16 fragment slots left a
      // wave fetches its first fragment twice (same bytes, same place) so every wave issues IPS instructions
      const int fi0 = wave + f * DS_NW, fi = fi0 < 16 ? fi0 : wave;
      ds_dma16(a.w + ((long long)(f_ks * 2 + (fi >> 3)) * DS_CT + (fi & 7)) * 512 + lane * 8, dst + DS_ACT_B + fi * 1024);
    }
    if (++f_ks == a.nks) {
      f_ks = 0;
      f_tile += (int)gridDim.x;
    }
  };

  // ---- consumer maps: B fragment of MFMA j for pixel tile p: local pixel 32 wave + 16 p + m, chunk 2g + j
  int boff[DS_PT][2];
#pragma unroll
  for (int p = 0; p < DS_PT; ++p)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int px = wave * C::WPX + p * 16 + m;
      boff[p][j] = px * 128 + (((2 * g + j) ^ ((px >> 1) & 5)) * 16);
    }
  // batch statistics of the lane's 32 outputs per pixel tile, as packed pairs (v_pk_add_f32 / v_pk_fma_f32: round 6 -- the epilogue was
  // two thirds of the kernel's VALU instructions, 128 of them these sums, and `conv1x1_ds` is issue-bound: its rate follows the CUs
  // it is given, profiles/r6_cu_mask_sweep.txt)
  // (Also tried in round 6, not kept: the tile's first stage starting from the MFMA's inline-zero accumulator operand instead of 64 v_mov
  // after the epilogue -- two copies of the stage body behind a uniform `ks == 0` cost 11 spilled registers and vmcnt(0) waits at the joins.)
  f32x2 st1[DS_CT][2], st2[DS_CT][2];
#pragma unroll
  for (int c = 0; c < DS_CT; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) st1[c][h] = st2[c][h] = f32x2{0.f, 0.f};
  f32x4 acc[DS_PT][DS_CT];
#pragma unroll
  for (int p = 0; p < DS_PT; ++p)
#pragma unroll
    for (int c = 0; c < DS_CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (total > 0) issue(0);
  if (total > 1) issue(1);
  int buf = 0, ks = 0, tile = (int)blockIdx.x, pend = 0;
  for (int s = 0; s < total; ++s) {
    // this wave's share of stage s has landed; the barrier adds everybody else's filter fragments and
    // retires the buffer consumed in stage s-1
    if (s + 1 < total) {
      if (pend > 0) {
        ds_wait_vm<DS_IPS + DS_NSTORE>();
        --pend;
      } else {
        ds_wait_vm<DS_IPS>();
      }
    } else {
      ds_wait_vm<0>();
    }
    tm.stamp(0);
    ds_barrier();
    tm.stamp(1);
    if (s + 2 < total) issue(buf >= 1 ? buf - 1 : DS_NBUF - 1);   // (buf + 2) % 3
    tm.stamp(2);
    const char* act = stage0 + buf * DS_STAGE_B;
    const char* wfrag = act + DS_ACT_B + lane * 16;
    const bool full = (unsigned)tile * DS_PX + DS_PX <= a.P;   // uniform
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cb = ks * 64 + g * 16 + j * 8;
      const bool cok = cb < cmax;
      f16x8 xf[DS_PT];
      u32x4 raw[DS_PT];
#pragma unroll
      for (int p = 0; p < DS_PT; ++p) raw[p] = lds_read16(act + boff[p][j]);
      if (XMODE != 0 && !DS_SKIP(4)) {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc_lds + cb), s1 = *reinterpret_cast<const f32x4*>(sc_lds + cb + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh_lds + cb), h1 = *reinterpret_cast<const f32x4*>(sh_lds + cb + 4);
#pragma unroll
        for (int p = 0; p < DS_PT; ++p) raw[p] = fd_xform8_r(raw[p], s0, s1, h0, h1, XMODE == 1 ? 0.f : a.p_slope);
      }
#pragma unroll
      for (int p = 0; p < DS_PT; ++p) {
        // a pixel past the end / a channel group past Cin must contribute exactly zero
        const bool ok = cok && (full || (unsigned)tile * DS_PX + wave * C::WPX + p * 16 + m < a.P);
        xf[p] = __builtin_bit_cast(f16x8, ok ? raw[p] : zero4);
      }
#pragma unroll
      for (int c0 = 0; c0 < DS_CT; c0 += 4) {
        f16x8 wf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) wf[c] = __builtin_bit_cast(f16x8, lds_read16(wfrag + (j * DS_CT + c0 + c) * 1024));
#pragma unroll
        for (int p = 0; p < DS_PT; ++p)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (!DS_SKIP(2)) acc[p][c0 + c] = fd_mfma_a(wf[c], xf[p], acc[p][c0 + c]);
      }
    }

    tm.stamp(3);
    if (ks + 1 == a.nks) {
      // ---- epilogue of `tile`: statistics (registers) and 256-byte row stores, staged through this
      // wave's own activation region of the buffer just consumed (nobody else touches it, and the
      // DMA that refills it is issued by this wave after the next barrier)
      char* tb = const_cast<char*>(act) + wave * (C::WPX * 128);
#pragma unroll
      for (int p = 0; p < DS_PT; ++p) {
        const unsigned pxt = (unsigned)tile * DS_PX + wave * C::WPX + p * 16;
        // whole tiles (every tile of the generator's shapes) skip the per-element pixel mask: a uniform branch with no load inside
        if (!DS_SKIP(32)) {
          if (full) {
#pragma unroll
            for (int c = 0; c < DS_CT; ++c)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const f32x2 v = {acc[p][c][2 * h], acc[p][c][2 * h + 1]};
                st1[c][h] += v;
                st2[c][h] = __builtin_elementwise_fma(v, v, st2[c][h]);
              }
          } else {
            const bool pok = pxt + m < a.P;
#pragma unroll
            for (int c = 0; c < DS_CT; ++c)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const f32x2 v = {pok ? acc[p][c][2 * h] : 0.f, pok ? acc[p][c][2 * h + 1] : 0.f};
                st1[c][h] += v;
                st2[c][h] = __builtin_elementwise_fma(v, v, st2[c][h]);
              }
          }
        }
        unsigned short* yrow = reinterpret_cast<unsigned short*>(a.y) + (unsigned long long)pxt * (unsigned)a.y_sw;
        const int npix = full ? 16 : (pxt < a.P ? (int)min(16u, a.P - pxt) : 0);
#pragma unroll
        for (int c0 = 0; c0 < DS_CT; c0 += C::SC) {
          float v[C::SC][4];
#pragma unroll
          for (int c = 0; c < C::SC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[c][r] = acc[p][c0 + c][r];
          if (!DS_SKIP(8)) fd_store_row16_ptr<C::SC>(yrow + c0 * 16, a.y_sw, tb, v, lane, npix);
        }
#pragma unroll
        for (int c = 0; c < DS_CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      pend = 2;
      ks = 0;
      tile += (int)gridDim.x;
      tm.stamp(4);
    } else {
      ++ks;
    }
    buf = buf + 1 == DS_NBUF ? 0 : buf + 1;
  }

  if (tm.on && lane == 0 && wave < 8)
    for (int k = 0; k < 6; ++k) a.dbg[wave * 8 + k] = tm.t[k];
  // ---- one partial row of statistics per workgroup
  if (a.stats != nullptr) {
#pragma unroll
    for (int c = 0; c < DS_CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = fd_row_sum16(st1[c][r >> 1][r & 1]), s2 = fd_row_sum16(st2[c][r >> 1][r & 1]);
        if (m == 0) {
          float* d = red + ((wave * 128) + c * 16 + g * 4 + r) * 2;
          d[0] = s1;
          d[1] = s2;
        }
      }
    __syncthreads();
    for (int cl = tid; cl < 128; cl += DS_NT) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < DS_NW; ++w_) {
        t1 += red[(w_ * 128 + cl) * 2];
        t2 += red[(w_ * 128 + cl) * 2 + 1];
      }
      float* dst = a.stats + ((long long)blockIdx.x * a.stats_cpad + cl) * 2;
      fd_stats_store(a, dst, t1);
      fd_stats_store(a, dst + 1, t2);
    }
    if (a.fin_mean != nullptr) fd_finalize_last_block(a, 128, tid, stage0);
  }
}

int ds_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int XMODE, int NW, int PT>
int ds_launch(const ConvArgs& a, dim3 grid, unsigned lds, hipStream_t stream) {
  auto kfn = &conv1x1_ds_kernel<XMODE, NW, PT>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv1x1_ds): %s", hipGetErrorString(e));
    attr_done = true;
  }
  return fd_launch(kfn, NW == 8 ? "conv1x1_ds_bn128" : (NW == 12 ? "conv1x1_ds_bn128_w12" : "conv1x1_ds_bn128_w16"), grid,
                   dim3(64 * NW, 1, 1), lds, a, stream);
}

}  // namespace

// Cout == 128 exactly, dense NHWC input / output, x64 filter layout, plain epilogue (no bias, no
// activation, no upsample), scale / shift of every input channel next to the three stage buffers.
bool conv1x1_ds_fits(const ConvArgs& a, int cout_total, bool pool, int w_layout) {
  return !pool && cout_total == 128 && a.ntile_total == 8 && w_layout == FD_WLAYOUT_X64 && a.x_dense && a.y_dense &&
         a.y_vec16 && a.Cout >= 128 && a.bias == nullptr && a.e_slope == 1.f && !a.upsample && !a.out_nchw_f32 &&
         (a.x_sw % 8) == 0 && a.Cin >= 8 && DsCfg<8, 2>::lds_bytes((a.Cin + 63) / 64) <= 160 * 1024;
}

template <int NW, int PT>
static int ds_dispatch(ConvArgs& a, FdConvInfo* info, long long stats_cap, bool dry, hipStream_t stream) {
  using C = DsCfg<NW, PT>;
  a.nks = (a.Cin + 63) / 64;
  a.ntiles = (int)((a.P + C::PX - 1) / C::PX);
  const int ncu = fd_cus(dry ? 256 : ds_num_cus());
  dim3 grid((unsigned)(a.ntiles < ncu ? a.ntiles : ncu), 1, 1);
  a.stats_cpad = 128;
  const unsigned lds = C::lds_bytes(a.nks);
  if (info) {
    info->stats_rows = grid.x;
    info->stats_cpad = a.stats_cpad;
    info->grid_x = grid.x;
    info->grid_y = 1;
    info->lds_bytes = lds;
    info->fused_finalize = 1;
  }
  if (dry) return FD_OK;
  if (stats_cap >= 0 && (long long)grid.x * a.stats_cpad * 2 > stats_cap)
    FD_FAIL(FD_EINVAL, "stats workspace too small: need %lld floats, have %lld", (long long)grid.x * a.stats_cpad * 2,
            stats_cap);
  if (a.pro_mode == 0) return ds_launch<0, NW, PT>(a, grid, lds, stream);
  if (a.p_slope == 0.f) return ds_launch<1, NW, PT>(a, grid, lds, stream);
  return ds_launch<2, NW, PT>(a, grid, lds, stream);
}

int conv_dispatch_k1_ds(ConvArgs& a, FdConvInfo* info, long long stats_cap, bool dry, hipStream_t stream) {
  // measured: 12 waves x 16 pixels (3 waves per SIMD) is 10-30 % SLOWER than 8 x 32 -- half the A-fragment
  // reuse, twice the LDS reads per MFMA -- and 16 waves do not fit LDS; one configuration is instantiated
  // (tuning aid, round 4: 8 x 16 pixels = 112 KB of LDS instead of 157, so that a 26-31 KB workgroup of the MFMA-bound kernels the
  // other stream runs beside the generator's forward -- D on the real batch, VGG16 on the target -- could share the CU.  Measured:
  // the forward alone 5.06 -> 5.43 ms, the training step 27.58 -> 27.83 ms: the slower kernel is not paid back.)
  static const char* small_env = FD_TUNE_GETENV("FDGAN_DEBUG_DS_SMALL");
  if (small_env && small_env[0] == '1') return ds_dispatch<8, 1>(a, info, stats_cap, dry, stream);
  return ds_dispatch<8, 2>(a, info, stats_cap, dry, stream);
}
