// conv3x3_pw.hip -- 3x3 stride-1 convolution with few output channels (<= 32) and <= 128
// input channels: the dense-layer growth conv 128 -> 32 (43 of netG's 49 3x3 convs, the
// HBM-bound headline shape) and the final 16 -> 3.  "pw" = persistent workgroups, filter
// resident, wave-specialised.
//
// Structure (one 8-wave workgroup per CU, walking 32x16-pixel tiles; a "step" is one
// 32-channel chunk of one tile):
//   * the whole filter (9 taps x nchunk x 32 couts, <= 72 KiB) is loaded into LDS once;
//   * waves 4-7 are STAGERS: they load the activation halo tile of a later step (34x18 pixels
//     x 32 channels; 16 pixels x 64 contiguous bytes per wave instruction), apply BatchNorm +
//     ReLU + zero padding, and write the MFMA-friendly LDS image (4 planes of [pixel][8 ch]);
//   * waves 0-3 are COMPUTE waves: 8 rows x 16 pixels x 32 couts each, 144 MFMAs per step fed
//     by 48 ds_read_b128, plus the epilogue (bias, activation, statistics, 64-byte row stores).
//   Measured on the un-specialised predecessor (s_memtime per phase + PMC): with all 8 waves
//   alternating "MFMA phase" and "staging phase" in lock-step behind the per-step barrier the
//   matrix pipe idled while the VALU staged and vice versa (10k cycles per step for 2.3k cycles
//   of MFMA).  Putting the two phases on different waves of the same SIMD lets them overlap.
//   * loads run two steps ahead in two register sets.  They are issued by inline asm with one
//     counted s_waitcnt per set: with compiler-visible loads hipcc waited vmcnt(4..0) where
//     vmcnt(9..5) was intended (verified in the ISA), collapsing the prefetch depth.
//   * batch statistics accumulate in registers across all tiles: one partial row per workgroup.
#include "conv_igemm.h"

// A set's loads are older than the other set's: loads return in order, so "<= N outstanding"
// with N = the younger set's load count means this set has landed; compiler-issued memory ops in
// between can only lengthen the wait.  The destination registers are read-write operands of the
// wait, so no use can be scheduled above it.
#define PW_ASM_LOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define PW_ASM_WAIT(n, r)                                                                                   \
  asm volatile("s_waitcnt vmcnt(" n ")"                                                                     \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                 "+v"(r[8]), "+v"(r[9])::"memory")

namespace {

constexpr int PW_NT = 512, PW_NCW = 4, PW_NSW = 4;               // compute / stager waves
constexpr int PW_PT = 8, PW_CT = 2;
constexpr int PW_TH = PW_NCW * PW_PT, PW_TW = 16;                // 32 x 16 output pixels
constexpr int PW_IH = PW_TH + 2, PW_IW = PW_TW + 2;              // 34 x 18 halo tile
constexpr int PW_NPIX = PW_IH * PW_IW;                           // 612
constexpr int PW_NPIXR = (PW_NPIX + 15) / 16 * 16;               // 624
constexpr int PW_NGRP = PW_NPIXR / 16;                           // 39 groups of 16 halo pixels
constexpr int PW_SUPT = (PW_NGRP + PW_NSW - 1) / PW_NSW;         // 10 groups per stager wave
constexpr int PW_PLANE_B = PW_NPIXR * 16;
constexpr int PW_UNITS = 4 * PW_NPIXR;
constexpr int PW_IN_BYTES = PW_UNITS * 16 + 1024;                // 4 planes + one wave-sized padding slot
constexpr int PW_MAXCHUNK = 4;
static_assert(PW_SUPT == 10, "PW_ASM_WAIT names 10 registers");

__host__ __device__ inline unsigned pw_w_bytes(int nchunk) { return (unsigned)nchunk * 9 * PW_CT * 1024; }
__host__ __device__ inline unsigned pw_lds_bytes(int nchunk) {
  return 2 * PW_IN_BYTES + pw_w_bytes(nchunk) + nchunk * 32 * 8 + PW_NCW * PW_CT * 16 * 2 * 4;
}

__global__ __launch_bounds__(PW_NT, 2) void conv3x3_pw_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* in_lds = smem;                                         // [2][IN_BYTES]
  char* w_lds = smem + 2 * PW_IN_BYTES;                        // [nchunk][9][CT][1 KiB]
  float* sc_lds = reinterpret_cast<float*>(w_lds + pw_w_bytes(a.nchunk));
  float* sh_lds = sc_lds + a.nchunk * 32;
  float* red = sh_lds + a.nchunk * 32;                         // [NCW][CT*16][2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably uniform role selector
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  fd_fold_bn(a, sc_lds, sh_lds, a.nchunk * 32, tid, PW_NT);
  {  // the whole filter -> LDS, once
    const int nunits = a.nchunk * 9 * PW_CT * 64;
    for (int u = tid; u < nunits; u += PW_NT) {
      const int kt = u / (PW_CT * 64), rem = u - kt * (PW_CT * 64);
      const int tile16 = rem >> 6;
      const bool ok = tile16 < a.ntile_total;
      const u32x4 v = *reinterpret_cast<const u32x4*>(
          a.w + (ok ? ((long long)kt * a.ntile_total + tile16) * 512 + (rem & 63) * 8 : 0));
      lds_write16(w_lds + u * 16, ok ? v : zero4);
    }
  }

  const int tiles_img = a.tiles_x * a.tiles_y;
  const int total_tiles = a.ntiles;
  const int my_tiles = blockIdx.x < total_tiles ? (total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int nsteps = my_tiles * a.nchunk;
  auto step_tile = [&](int k) { return (int)blockIdx.x + (k / a.nchunk) * (int)gridDim.x; };
  auto step_chunk = [&](int k) { return k % a.nchunk; };

  __syncthreads();   // scale/shift + filter visible

  // Both roles execute the same barrier sequence: one after the prologue staging, one per step,
  // one more after each tile's epilogue, and the final one before the statistics reduction.
  if (wave >= PW_NCW) {
    // =========================== STAGER waves ===========================
    // lane = pixel*4 + channel group on the way in (4 lanes read one pixel's 64 contiguous bytes);
    // 4 ds_bpermute_b32 per unit move it to lane' = group*16 + pixel, whose 16 consecutive lanes
    // write 256 contiguous bytes of one LDS plane (conflict-free) and share ONE channel group, so
    // the BN scale/shift is fetched once per step.
    const int sw = wave - PW_NCW;
    const int sq = lane >> 2, skg = lane & 3;
    const int dq = lane & 15, dkg = lane >> 4;
    const int perm_addr = (dq * 4 + dkg) * 4;
    int off_src[PW_SUPT], lds_off[PW_SUPT];
#pragma unroll
    for (int i = 0; i < PW_SUPT; ++i) {
      const int grp = i * PW_NSW + sw;
      const bool gex = grp < PW_NGRP;                              // uniform per wave
      const int ps = grp * 16 + sq, pd = grp * 16 + dq;
      const int spy = ps / PW_IW, spx = ps - spy * PW_IW;
      off_src[i] = (gex && ps < PW_NPIX) ? (spy - 1) * a.x_sh + (spx - 1) * a.x_sw + skg * 8 : 0;
      lds_off[i] = gex ? dkg * PW_PLANE_B + pd * 16 : PW_UNITS * 16 + lane * 16;   // padding slot
    }
    constexpr unsigned ALL_UNITS = (1u << PW_SUPT) - 1;
    // Two register sets: loads run two steps ahead (20 KiB per stager wave, 80 KiB per CU in flight).
    // The 2-D halo-tile read pattern has ~4.6 us loaded latency on MI355X (tools/ubench/tile2d.hip:
    // 2.2 TB/s at 40 KiB in flight per CU, 3.9 TB/s at 160 KiB), so the achieved bandwidth is
    // bytes-in-flight / latency; a third set (120 KiB) spilled registers and ran 2.5x slower.
    u32x4 rinA[PW_SUPT], rinB[PW_SUPT];
    unsigned lmaskA = 0, lmaskB = 0;   // bit i: this lane's DESTINATION unit i holds real data (else zeros)
    auto load_in = [&](u32x4(&rin)[PW_SUPT], unsigned& lmask, int tile, int chunk) {
      const int n = tile / tiles_img, t2 = tile - n * tiles_img;
      const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
      const unsigned short* base = a.x + (long long)n * a.x_sn + (long long)(ty * PW_TH) * a.x_sh +
                                   (long long)(tx * PW_TW) * a.x_sw + chunk * 32;
      const bool interior = ty > 0 && (ty + 1) * PW_TH < a.Hs && tx > 0 && (tx + 1) * PW_TW < a.Ws &&
                            chunk * 4 + 3 < a.Cin8;   // uniform: no bounds tests, no zeroing
      if (interior) {
        lmask = ALL_UNITS;
#pragma unroll
        for (int i = 0; i < PW_SUPT; ++i) PW_ASM_LOAD(rin[i], base + off_src[i]);
      } else {
        lmask = 0;
#pragma unroll
        for (int i = 0; i < PW_SUPT; ++i) {
          const int grp = i * PW_NSW + sw;
          const int ps = grp * 16 + sq, pd = grp * 16 + dq;
          const int spy = ps / PW_IW, spx = ps - spy * PW_IW;
          const int sgy = ty * PW_TH - 1 + spy, sgx = tx * PW_TW - 1 + spx;
          const bool sok = sgy >= 0 && sgy < a.Hs && sgx >= 0 && sgx < a.Ws && (chunk * 4 + skg) < a.Cin8;
          PW_ASM_LOAD(rin[i], base + (sok ? off_src[i] : 0));   // clamped to the tile origin
          const int dpy = pd / PW_IW, dpx = pd - dpy * PW_IW;
          const int gy = ty * PW_TH - 1 + dpy, gx = tx * PW_TW - 1 + dpx;
          const bool dok = gy >= 0 && gy < a.Hs && gx >= 0 && gx < a.Ws && (chunk * 4 + dkg) < a.Cin8;
          lmask |= dok ? (1u << i) : 0u;
        }
      }
    };
    auto store_in = [&](u32x4(&rin)[PW_SUPT], unsigned lmask, char* buf, int chunk, int younger_sets) {
      if (younger_sets >= 2)
        PW_ASM_WAIT("20", rin);
      else if (younger_sets == 1)
        PW_ASM_WAIT("10", rin);
      else
        PW_ASM_WAIT("0", rin);
      const bool all = __all((int)(lmask == ALL_UNITS));
      const float* sc = sc_lds + chunk * 32 + dkg * 8;
      const float* sh = sh_lds + chunk * 32 + dkg * 8;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh), h1 = *reinterpret_cast<const f32x4*>(sh + 4);
#pragma unroll
      for (int i = 0; i < PW_SUPT; ++i)
#pragma unroll
        for (int d = 0; d < 4; ++d) rin[i][d] = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)rin[i][d]);
      if (a.pro_mode != 0) {
        if (a.p_slope == 0.f) {
#pragma unroll
          for (int i = 0; i < PW_SUPT; ++i) rin[i] = fd_xform8_r(rin[i], s0, s1, h0, h1, 0.f);
        } else {
#pragma unroll
          for (int i = 0; i < PW_SUPT; ++i) rin[i] = fd_xform8_r(rin[i], s0, s1, h0, h1, a.p_slope);
        }
      }
      if (all) {
#pragma unroll
        for (int i = 0; i < PW_SUPT; ++i) lds_write16(buf + lds_off[i], rin[i]);
      } else {
#pragma unroll
        for (int i = 0; i < PW_SUPT; ++i)
          lds_write16(buf + lds_off[i], ((lmask >> i) & 1u) ? rin[i] : zero4);   // zero padding is post-activation
      }
    };

    if (nsteps > 0) {
      load_in(rinA, lmaskA, step_tile(0), 0);
      store_in(rinA, lmaskA, in_lds, 0, 0);
      if (nsteps > 1) load_in(rinB, lmaskB, step_tile(1), step_chunk(1));
      if (nsteps > 2) load_in(rinA, lmaskA, step_tile(2), step_chunk(2));
    }
    __syncthreads();   // first step staged
    auto sstep = [&](int s, u32x4(&rnext)[PW_SUPT], unsigned& lnext) {
      if (s + 1 < nsteps)
        store_in(rnext, lnext, in_lds + ((s + 1) & 1) * PW_IN_BYTES, step_chunk(s + 1),
                 s + 2 < nsteps ? 1 : 0);
      if (s + 3 < nsteps) load_in(rnext, lnext, step_tile(s + 3), step_chunk(s + 3));
      __syncthreads();
      if (step_chunk(s) + 1 == a.nchunk) __syncthreads();   // the compute waves' epilogue owns buffer s&1 until here
    };
    for (int s = 0; s < nsteps; s += 2) {   // step s stages step s+1, carried by set (s+1)&1
      sstep(s, rinB, lmaskB);
      if (s + 1 < nsteps) sstep(s + 1, rinA, lmaskA);
    }
  } else {
    // =========================== COMPUTE waves ===========================
    const int m = lane & 15, kgl = lane >> 4;
    f32x4 acc[PW_PT][PW_CT];
    float st1[PW_CT][4], st2[PW_CT][4];   // statistics, summed over every tile of this workgroup
#pragma unroll
    for (int c = 0; c < PW_CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) st1[c][r] = st2[c][r] = 0.f;
#pragma unroll
    for (int p = 0; p < PW_PT; ++p)
#pragma unroll
      for (int c = 0; c < PW_CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();   // first step staged

    const char* xfrag0 = in_lds + kgl * PW_PLANE_B + ((wave * PW_PT) * PW_IW + m) * 16;
    const char* wfrag0 = w_lds + lane * 16;
    const bool rowstore_ok = a.y_vec16 && PW_CT * 16 <= a.Cout;

    for (int s = 0; s < nsteps; ++s) {
      const int tile = step_tile(s), chunk = step_chunk(s);
      const char* xb = xfrag0 + (s & 1) * PW_IN_BYTES;
      const char* wb = wfrag0 + chunk * (9 * PW_CT * 1024);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        // the 10 input rows of this wave, read once per column shift and shared by the three row taps
        f16x8 xr[PW_PT + 2];
#pragma unroll
        for (int r = 0; r < PW_PT + 2; ++r)
          xr[r] = __builtin_bit_cast(f16x8, lds_read16(xb + (r * PW_IW + dx) * 16));
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          f16x8 wf[PW_CT];
#pragma unroll
          for (int c = 0; c < PW_CT; ++c)
            wf[c] = __builtin_bit_cast(f16x8, lds_read16(wb + ((dy * 3 + dx) * PW_CT + c) * 1024));
#pragma unroll
          for (int p = 0; p < PW_PT; ++p)
#pragma unroll
            for (int c = 0; c < PW_CT; ++c)
              acc[p][c] = fd_mfma_a(wf[c], xr[p + dy], acc[p][c]);
        }
      }
      __syncthreads();

      if (chunk + 1 == a.nchunk) {
        // ---- epilogue of `tile`: bias, activation, row stores, statistics.  The buffer just consumed
        // (s & 1) is free until the stagers pass the barrier below: use it as row-store staging.
        const int n = tile / tiles_img, t2 = tile - n * tiles_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int oy0 = ty * PW_TH, ox0 = tx * PW_TW, col = ox0 + m;
        char* tb = in_lds + (s & 1) * PW_IN_BYTES + wave * RowStore<PW_CT>::BYTES;
        float bv[PW_CT][4];
#pragma unroll
        for (int c = 0; c < PW_CT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = c * 16 + kgl * 4 + r;
            bv[c][r] = (a.bias != nullptr && co < a.CoutW) ? a.bias[co] : 0.f;
          }
#pragma unroll
        for (int p = 0; p < PW_PT; ++p) {
          const int row = oy0 + wave * PW_PT + p;
          const bool valid = row < a.Ho && col < a.Wo;
          float v[PW_CT][4];
#pragma unroll
          for (int c = 0; c < PW_CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float t = acc[p][c][r] + bv[c][r];
              v[c][r] = fmaxf(t, a.e_slope * t);
              st1[c][r] += valid ? v[c][r] : 0.f;
              st2[c][r] += valid ? v[c][r] * v[c][r] : 0.f;
              acc[p][c][r] = 0.f;
            }
          if (rowstore_ok) {
            fd_store_row16<PW_CT>(a, tb, v, lane, 0, [&](int q) -> long long {
              return (row < a.Ho && ox0 + q < a.Wo)
                         ? (long long)n * a.y_sn + (long long)row * a.y_sh + (long long)(ox0 + q) * a.y_sw
                         : -1;
            });
          } else if (valid) {
            const int up = a.upsample ? 2 : 1;
            const long long off =
                (long long)n * a.y_sn + (long long)(up * row) * a.y_sh + (long long)(up * col) * a.y_sw;
#pragma unroll
            for (int c = 0; c < PW_CT; ++c)
              if (c * 16 + kgl * 4 < a.Cout) fd_store4(a, off, c * 16 + kgl * 4, v[c]);
          }
        }
        __syncthreads();   // hand buffer s&1 back to the stagers
      }
    }

    if (a.stats != nullptr) {
#pragma unroll
      for (int c = 0; c < PW_CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = fd_row_sum16(st1[c][r]), s2 = fd_row_sum16(st2[c][r]);
          if (m == 0) {
            const int idx = (wave * PW_CT * 16 + c * 16 + kgl * 4 + r) * 2;
            red[idx] = s1;
            red[idx + 1] = s2;
          }
        }
    }
  }

  // ---- common tail: one partial row of statistics per workgroup
  if (a.stats != nullptr) {
    __syncthreads();
    for (int cl = tid; cl < PW_CT * 16; cl += PW_NT) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < PW_NCW; ++w_) {
        t1 += red[(w_ * PW_CT * 16 + cl) * 2];
        t2 += red[(w_ * PW_CT * 16 + cl) * 2 + 1];
      }
      float* dst = a.stats + ((long long)blockIdx.x * a.stats_cpad + cl) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

int pw_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

}  // namespace

bool conv3x3_pw_fits(int cout_total, int cin) { return cout_total <= 32 && (cin + 31) / 32 <= PW_MAXCHUNK; }

int conv_dispatch_k3_pw(ConvArgs& a, long long nimg, int cout_total, FdConvInfo* info, long long stats_cap, bool dry,
                        hipStream_t stream) {
  if (!conv3x3_pw_fits(cout_total, a.Cin)) FD_FAIL(FD_EUNSUPPORTED, "conv3x3_pw: shape not supported");
  a.tiles_x = (a.Wo + PW_TW - 1) / PW_TW;
  a.tiles_y = (a.Ho + PW_TH - 1) / PW_TH;
  const long long nt = nimg * a.tiles_x * a.tiles_y;
  if (nt >= (1ll << 31)) FD_FAIL(FD_EUNSUPPORTED, "conv3x3_pw: too many tiles");
  a.ntiles = (int)nt;
  const int ncu = fd_cus(dry ? 256 : pw_num_cus());
  dim3 grid((unsigned)(nt < ncu ? nt : ncu), 1, 1), block(PW_NT, 1, 1);
  a.stats_cpad = PW_CT * 16;
  const unsigned lds = pw_lds_bytes(a.nchunk);
  if (info) {
    info->stats_rows = grid.x;
    info->stats_cpad = a.stats_cpad;
    info->grid_x = grid.x;
    info->grid_y = 1;
    info->lds_bytes = lds;
  }
  if (dry) return FD_OK;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pw_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv3x3_pw): %s", hipGetErrorString(e));
    attr_done = true;
  }
  if (stats_cap >= 0 && (long long)grid.x * a.stats_cpad * 2 > stats_cap)
    FD_FAIL(FD_EINVAL, "stats workspace too small: need %lld floats, have %lld", (long long)grid.x * a.stats_cpad * 2,
            stats_cap);
  return fd_launch(&conv3x3_pw_kernel, "conv3x3_pw_bn32", grid, block, lds, a, stream);
}
