// conv3x3_rs.hip -- the dense-layer growth convolution: 3x3, stride 1, pad 1, exactly 128 input
// channels, <= 32 output channels (42 of netG's 49 3x3 convs; the HBM-bound headline shape
// 128 -> 32 @ 256x256, B=16: 268 MB in, 67 MB out, 77 GFLOP).  "rs" = row streaming.
//
// Work item = a 16-pixel-wide column strip of one image (optionally cut into row segments so that
// every CU gets an item); one 12-wave workgroup walks down its strip four output rows at a time.
// Three wave roles, one wave of each role per SIMD:
//
//   HBM -> LDS   Input rows of the strip (18 pixels x 256 B, contiguous in memory) are fetched by
//                LDS-DMA (global_load_lds_dwordx4: no staging registers) into a ring of NR rows, PF
//                row groups ahead.  Measured (tools/ubench/strip.hip): this access pattern streams
//                at 5.3-5.7 TB/s useful with as little as 18 KiB in flight per CU, against 2.2-3.9 TB/s
//                for the 2-D halo tiles of 64-byte pieces its predecessor (conv3x3_pw) reads; and
//                rocprofv3 FETCH_SIZE x2 = 269 MB per launch for 268 MB of input (the 12.5 % column
//                halo is served by the L2 of the XCD that owns the neighbouring strip).
//   LDS layout   row = [18 px][16 slots of 16 B]; slot = chunk16 ^ (2*px & 15), applied on the DMA
//                SOURCE address (the LDS side of the DMA is lane-linear).  With this swizzle the
//                ds_read_b128 of an MFMA B fragment (16 pixels x one 16-byte chunk, for all three
//                column shifts) hits 16 distinct slots in each of the instruction's lane groups.
//   waves 0-3    COMPUTE, split along K: wave w owns input channels [32w, 32w+32) and keeps its
//                quarter of the filter -- 9 taps x 2 cout tiles = 18 A fragments -- in 72 VGPRs for
//                the whole launch (no filter in LDS, no A-operand reads).  Per iteration: 18
//                ds_read_b128 (6 rows x 3 column shifts) feed 72 MFMAs (4 rows x 9 taps x 2 tiles);
//                the 8 accumulator fragments go to LDS as partial sums.  Nothing else: the matrix
//                pipe of each SIMD is fed by a wave that does no prologue or epilogue work.
//   waves 4-7    LOADERS.  Loader h issues the DMA of its row of a group and, one group later,
//                applies BatchNorm + activation + zero padding to that row IN PLACE (5 ds_read_b128
//                -> 20 VALU each -> 5 ds_write_b128).  A row is transformed by the wave that
//                fetched it, so its own s_waitcnt vmcnt is the only synchronisation between the two.
//   waves 8-11   FINISHERS.  Finisher h runs the epilogue of output row h of the PREVIOUS iteration:
//                adds the four K-partials, bias, activation, statistics, one 64-byte store per pixel.
//   per iteration one s_barrier: it publishes the next group's transformed rows and this
//   iteration's partial sums, and retires the four oldest ring rows.
//
// How it got here (s_memtime per phase, FDGAN_TIMING=1 in tools/conv_bench.py; 128->32 @256^2 B=16):
//   651 us  first version, epilogue on the compute waves, lambdas with uniform branches inside
//   145 us  same, after hipcc stopped serialising the MFMA loop (uniform skip flags changed scheduling)
//   135 us  epilogue moved to 4 helper waves; role bodies templated so the unit loops are branch-free
//           (with the branches every ds_read sat behind its own s_waitcnt: 2400 cycles per row transform)
//   117 us  helpers split into loaders and finishers (12 waves): a wave issues at most one instruction
//           per 4 cycles, and a helper that fetched, transformed AND finished needed ~700 of them
// against 179 us for conv3x3_pw.  An iteration still takes ~3100 cycles for 1152 cycles of MFMA: LDS
// (~1300 cycles of traffic per iteration, 40 % of it the K-partial exchange) and the VALU issue slots
// shared by the three waves of a SIMD (~330 VALU + 72 MFMA per iteration) are the next limits.
#include "conv_igemm.h"

namespace {

constexpr int RS_NT = 768, RS_TW = 16, RS_IW = RS_TW + 2;
constexpr int RS_ROW_B = RS_IW * 256;         // 4608: one ring row
constexpr int RS_R = 4;                       // output rows per iteration = waves per role
constexpr int RS_NR = 18, RS_PF = 3;          // ring rows; groups fetched ahead
constexpr int RS_CT = 2;
constexpr int RS_RING_B = RS_NR * RS_ROW_B;
constexpr int RS_RED_B = RS_R * RS_R * RS_CT * 1024;   // partial sums of one iteration: [row][k-quarter][ct]
// group i+PF (rows 4(i+PF)+2 .. +5) is fetched while rows 4i .. 4(i+PF)+1 are live
static_assert(RS_NR >= 4 * RS_PF + 6, "ring too small for the prefetch depth");

__host__ __device__ inline unsigned rs_lds_bytes() {
  return RS_RING_B + 2 * RS_RED_B + 128 * 8 + RS_R * 32 * 2 * 4 + RS_R * RowStore<RS_CT>::BYTES;
}

__device__ __forceinline__ void rs_dma16(const unsigned short* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void rs_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Workgroup barrier that leaves LDS-DMA in flight: __syncthreads() carries a workgroup release, which
// hipcc lowers to s_waitcnt vmcnt(0) while a global_load_lds is outstanding and would drain the
// prefetch at every iteration.  Only LDS traffic has to be complete here.
__device__ __forceinline__ void rs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Wait until at most `groups` younger 5-instruction row fetches of this wave are outstanding (memory
// operations of a wave retire in order; the helpers' output stores only make the wait conservative).
__device__ __forceinline__ void rs_wait_groups(int groups) {
  if (groups >= 2)
    rs_wait_vm<10>();
  else if (groups == 1)
    rs_wait_vm<5>();
  else
    rs_wait_vm<0>();
}

struct RsTimer {   // measurement aid (tools/conv_bench.py FDGAN_TIMING=1): s_memtime per phase, workgroup 0
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

struct RsItem {
  int n, sg, sx, n_iter;
};
__device__ __forceinline__ RsItem rs_item(const ConvArgs& a, int item) {
  RsItem it;
  const int items_img = a.tiles_x * a.tiles_y;
  it.n = item / items_img;
  const int r2 = item - it.n * items_img;
  it.sg = r2 / a.tiles_x;
  it.sx = r2 - it.sg * a.tiles_x;
  const int rows_here = min(a.seg_rows, a.Ho - it.sg * a.seg_rows);
  it.n_iter = (rows_here + RS_R - 1) / RS_R;
  return it;
}

// XMODE: 0 raw input (only the zero padding is written), 1 BatchNorm + ReLU, 2 affine + max(v, slope*v)
template <int XMODE, bool MASK>
__device__ __forceinline__ void rs_xform_row(char* row, const int (&xf_off)[5], bool rowok, unsigned colmask, int tp,
                                             int lane, f32x4 s0, f32x4 s1, f32x4 h0, f32x4 h1, float slope) {
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const bool upper = lane < 32;   // unit 4 exists for pixels 16, 17 only
  if (XMODE == 0) {
    if (!MASK) return;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const bool ok = rowok && ((colmask >> (it * 4 + tp)) & 1u);
      if (!ok && (it < 4 || upper)) lds_write16(row + xf_off[it], zero4);
    }
    return;
  }
  u32x4 v[5];
#pragma unroll
  for (int it = 0; it < 4; ++it) v[it] = lds_read16(row + xf_off[it]);
  v[4] = zero4;
  if (upper) v[4] = lds_read16(row + xf_off[4]);
#pragma unroll
  for (int it = 0; it < 5; ++it) {
    const u32x4 t = fd_xform8_r(v[it], s0, s1, h0, h1, XMODE == 1 ? 0.f : slope);
    if (MASK) {
      const bool ok = rowok && ((colmask >> (it * 4 + tp)) & 1u);
      v[it] = ok ? t : zero4;   // zero padding is post-activation
    } else {
      v[it] = t;
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) lds_write16(row + xf_off[it], v[it]);
  if (upper) lds_write16(row + xf_off[4], v[4]);
}

// ======================================= LOADER waves =======================================
// Loader h fetches input row 4g + 2 + h of every group g (rows h and 4 + h of group 0) and, one
// group later, transforms it in place.
template <int XMODE>
__device__ __forceinline__ void rs_loader(const ConvArgs& a, char* ring, const float* sc_lds, const float* sh_lds, int h,
                                          int lane, int bid, int nwg, RsTimer& tm) {
  const int tc = lane & 15, tp = lane >> 4;   // transform: 16-byte chunk, pixel within a group of 4
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, h0 = s0, h1 = s0;   // scale / shift of this lane's 8 channels (after barrier F)
  bool first = true;
  int xf_off[5];   // transform: byte offset of this lane's unit `it` inside a ring row
#pragma unroll
  for (int it = 0; it < 5; ++it) {
    const int p = it * 4 + tp;
    xf_off[it] = p * 256 + ((tc ^ ((2 * p) & 15)) * 16);
  }
  const unsigned row_pitch_b = (unsigned)a.x_sh * 2u;

  for (int item = bid; item < a.ntiles; item += nwg) {
    const RsItem I = rs_item(a, item);
    const int n_iter = I.n_iter;
    const int x0 = I.sx * RS_TW - 1, y0 = I.sg * a.seg_rows - 1;   // image coords of ring pixel 0 / ring row 0
    const char* img = reinterpret_cast<const char*>(a.x + (long long)I.n * a.x_sn);
    // DMA source: lane -> (pixel it*4 + lane/16, LDS slot lane%16) -> channel chunk slot ^ (2p & 15);
    // columns outside the image are fetched from the clamped column and zeroed by the transform.
    // Uniform 64-bit row base + per-lane 32-bit byte offset: the scalar-base form of the instruction.
    unsigned src_off[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int p = it * 4 + (lane >> 4), slot = lane & 15;
      const int gx = min(max(x0 + p, 0), a.Ws - 1);
      src_off[it] = (unsigned)(gx * a.x_sw + ((slot ^ ((2 * p) & 15)) * 8)) * 2u;
    }
    unsigned colmask = 0;   // bit p: ring pixel p lies inside the image
#pragma unroll
    for (int p = 0; p < RS_IW; ++p) colmask |= (x0 + p >= 0 && x0 + p < a.Ws) ? (1u << p) : 0u;
    const bool cols_in = colmask == (1u << RS_IW) - 1u;

    auto issue_row = [&](int q, int slot) __attribute__((always_inline)) {
      const int gy = min(max(y0 + q, 0), a.Hs - 1);
      const char* rowp = img + (unsigned long long)gy * row_pitch_b;
      char* dst = ring + slot * RS_ROW_B;
      if (a.dbg_skip & 1) return;
#pragma unroll
      for (int it = 0; it < 4; ++it) rs_dma16(reinterpret_cast<const unsigned short*>(rowp + src_off[it]), dst + it * 1024);
      if (lane < 32) rs_dma16(reinterpret_cast<const unsigned short*>(rowp + src_off[4]), dst + 4 * 1024);
    };
    auto xform_row = [&](int q, int slot) __attribute__((always_inline)) {
      if (a.dbg_skip & 2) return;
      const bool rowok = (y0 + q >= 0) && (y0 + q < a.Hs);
      char* row = ring + slot * RS_ROW_B;
      if (rowok && cols_in)   // uniform: nothing to zero
        rs_xform_row<XMODE, false>(row, xf_off, true, colmask, tp, lane, s0, s1, h0, h1, a.p_slope);
      else
        rs_xform_row<XMODE, true>(row, xf_off, rowok, colmask, tp, lane, s0, s1, h0, h1, a.p_slope);
    };
    // ring slots advance by 4 per group: rows 4g+2+h -> slot (4g + 2 + h) mod NR
    auto slot_of = [&](int q) __attribute__((always_inline)) { return q % RS_NR; };

    tm.stamp(5);
    // group 0: the 6 rows of iteration 0
    issue_row(h, h);
    if (h < 2) issue_row(4 + h, 4 + h);
#pragma unroll
    for (int gi = 1; gi < RS_PF; ++gi)
      if (gi < n_iter) issue_row(4 * gi + 2 + h, slot_of(4 * gi + 2 + h));
    tm.stamp(0);
    if (first) {   // barrier F: the other waves have folded BatchNorm into (scale, shift) meanwhile
      first = false;
      rs_barrier();
      s0 = *reinterpret_cast<const f32x4*>(sc_lds + tc * 8);
      s1 = *reinterpret_cast<const f32x4*>(sc_lds + tc * 8 + 4);
      h0 = *reinterpret_cast<const f32x4*>(sh_lds + tc * 8);
      h1 = *reinterpret_cast<const f32x4*>(sh_lds + tc * 8 + 4);
    }
    rs_wait_groups(min(RS_PF - 1, n_iter - 1));
    tm.stamp(1);
    xform_row(h, h);
    if (h < 2) xform_row(4 + h, 4 + h);
    tm.stamp(2);
    rs_barrier();   // P: iteration 0 may start
    tm.stamp(3);
    int q_x = 6 + h, s_x = 6 + h;                                  // row / slot transformed in iteration 0 (group 1)
    int q_i = 4 * RS_PF + 2 + h, s_i = (4 * RS_PF + 2 + h) % RS_NR;   // row / slot fetched in iteration 0 (group PF)
    for (int i = 0; i < n_iter; ++i) {
      if (i + RS_PF < n_iter) issue_row(q_i, s_i);
      tm.stamp(0);
      if (i + 1 < n_iter) {
        rs_wait_groups(min(i + RS_PF, n_iter - 1) - (i + 1));
        tm.stamp(1);
        xform_row(q_x, s_x);
        tm.stamp(2);
      }
      q_x += 4;
      q_i += 4;
      s_x = s_x + 4 >= RS_NR ? s_x + 4 - RS_NR : s_x + 4;
      s_i = s_i + 4 >= RS_NR ? s_i + 4 - RS_NR : s_i + 4;
      rs_barrier();   // iteration i retired: group i+1 published
      tm.stamp(3);
    }
  }
}

// ======================================= FINISHER waves =======================================
// Finisher h owns output row 4i + h of every iteration: sum of the four K-partials, bias, activation,
// statistics, one 64-byte store per pixel.  It runs one iteration behind the compute waves.
__device__ __forceinline__ void rs_finisher(const ConvArgs& a, const char* red, float* stat_red, char* rowstage, int h,
                                            int lane, int bid, int nwg, RsTimer& tm) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  const int m = lane & 15, g = lane >> 4;   // MFMA result layout: pixel m, couts g*4.. of tile c
  f32x2_t bv[RS_CT][2], st1[RS_CT][2], st2[RS_CT][2];
#pragma unroll
  for (int c = 0; c < RS_CT; ++c)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int co = c * 16 + g * 4 + 2 * r;
      bv[c][r][0] = (a.bias != nullptr && co < a.CoutW) ? a.bias[co] : 0.f;
      bv[c][r][1] = (a.bias != nullptr && co + 1 < a.CoutW) ? a.bias[co + 1] : 0.f;
      st1[c][r] = st2[c][r] = (f32x2_t){0.f, 0.f};
    }
  const bool rowstore_ok = a.y_vec16 && RS_CT * 16 <= a.Cout;
  const bool plain = a.e_slope == 1.f;   // no epilogue activation (the growth convs)
  char* tb = rowstage + h * RowStore<RS_CT>::BYTES;
  int parity = 0;
  rs_barrier();   // F (see the kernel body)

  for (int item = bid; item < a.ntiles; item += nwg) {
    const RsItem I = rs_item(a, item);
    const int n_iter = I.n_iter;
    const int ox0 = I.sx * RS_TW;
    const bool cols_full = ox0 + RS_TW <= a.Wo;
    const bool colvalid = ox0 + m < a.Wo;
    int row = I.sg * a.seg_rows + h;
    unsigned short* yrow = reinterpret_cast<unsigned short*>(a.y) + (long long)I.n * a.y_sn + (long long)ox0 * a.y_sw +
                           (long long)row * a.y_sh;   // uniform; advanced by 4 rows per iteration

    auto epilogue = [&]() __attribute__((always_inline)) {
      const char* rbuf = red + parity * RS_RED_B + h * (RS_R * RS_CT * 1024) + lane * 16;
      parity ^= 1;
      if ((a.dbg_skip & 8) || row >= a.Ho) return;   // uniform
      f32x4 part[RS_R][RS_CT];
#pragma unroll
      for (int k = 0; k < RS_R; ++k)
#pragma unroll
        for (int c = 0; c < RS_CT; ++c) part[k][c] = *reinterpret_cast<const f32x4*>(rbuf + (k * RS_CT + c) * 1024);
      float v[RS_CT][4];
#pragma unroll
      for (int c = 0; c < RS_CT; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          f32x2_t t = ((f32x2_t){part[0][c][2 * r], part[0][c][2 * r + 1]} + (f32x2_t){part[1][c][2 * r], part[1][c][2 * r + 1]}) +
                      ((f32x2_t){part[2][c][2 * r], part[2][c][2 * r + 1]} + (f32x2_t){part[3][c][2 * r], part[3][c][2 * r + 1]});
          t += bv[c][r];
          if (!plain) t = __builtin_elementwise_max(t, t * a.e_slope);
          const f32x2_t tv = (cols_full || colvalid) ? t : (f32x2_t){0.f, 0.f};
          st1[c][r] += tv;
          st2[c][r] = __builtin_elementwise_fma(tv, tv, st2[c][r]);
          v[c][2 * r] = t[0];
          v[c][2 * r + 1] = t[1];
        }
      if (a.dbg_skip & 16) return;
      if (rowstore_ok) {
        fd_store_row16_ptr<RS_CT>(yrow, a.y_sw, tb, v, lane, cols_full ? 16 : a.Wo - ox0);
      } else if (colvalid) {
        const int up = a.upsample ? 2 : 1;
        const long long off =
            (long long)I.n * a.y_sn + (long long)(up * row) * a.y_sh + (long long)(up * (ox0 + m)) * a.y_sw;
#pragma unroll
        for (int c = 0; c < RS_CT; ++c)
          if (c * 16 + g * 4 < a.Cout) fd_store4(a, off, c * 16 + g * 4, v[c]);
      }
    };

    tm.stamp(5);
    rs_barrier();   // P
    tm.stamp(3);
    for (int i = 0; i < n_iter; ++i) {
      if (i > 0) {
        epilogue();
        row += RS_R;
        yrow += (long long)RS_R * a.y_sh;
      }
      tm.stamp(4);
      rs_barrier();   // partial sums of iteration i published
      tm.stamp(3);
    }
    epilogue();
    tm.stamp(4);
  }
  if (a.stats != nullptr) {
#pragma unroll
    for (int c = 0; c < RS_CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1v = fd_row_sum16(st1[c][r >> 1][r & 1]), s2v = fd_row_sum16(st2[c][r >> 1][r & 1]);
        if (m == 0) {
          const int idx = (h * RS_CT * 16 + c * 16 + g * 4 + r) * 2;
          stat_red[idx] = s1v;
          stat_red[idx + 1] = s2v;
        }
      }
  }
}

// ======================================= COMPUTE waves =======================================
__device__ __forceinline__ void rs_compute(const ConvArgs& a, const char* ring, char* red, int w, int lane, int bid,
                                           int nwg, RsTimer& tm) {
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int m = lane & 15, g = lane >> 4;
  bf16x8 wf[9][RS_CT];   // this wave's quarter of the filter: channels [32w, 32w+32), all taps
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < RS_CT; ++c) {
      const bool ok = c < a.ntile_total;
      const u32x4 v = *reinterpret_cast<const u32x4*>(
          a.w + (ok ? ((long long)(w * 9 + t) * a.ntile_total + c) * 512 + lane * 8 : 0));
      wf[t][c] = __builtin_bit_cast(bf16x8, ok ? v : zero4);
    }
  int boff[3];   // B fragment: pixel m + dx, 16-byte chunk 4w + g
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) boff[dx] = (m + dx) * 256 + (((4 * w + g) ^ ((2 * (m + dx)) & 15)) * 16);
  int parity = 0;
  const f32x4 fzero = {0.f, 0.f, 0.f, 0.f};
  rs_barrier();   // F (see the kernel body)

  for (int item = bid; item < a.ntiles; item += nwg) {
    const int n_iter = rs_item(a, item).n_iter;
    tm.stamp(5);
    rs_barrier();   // P
    tm.stamp(2);
    int slot0 = 0;   // ring slot of input row 4i
    for (int i = 0; i < n_iter; ++i) {
      const char* rb[RS_R + 2];
#pragma unroll
      for (int r = 0; r < RS_R + 2; ++r) {
        const int sl = slot0 + r >= RS_NR ? slot0 + r - RS_NR : slot0 + r;
        rb[r] = ring + sl * RS_ROW_B;
      }
      slot0 = slot0 + RS_R >= RS_NR ? slot0 + RS_R - RS_NR : slot0 + RS_R;
      f32x4 acc[RS_R][RS_CT];
      if (!(a.dbg_skip & 4)) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          bf16x8 xr[RS_R + 2];
#pragma unroll
          for (int r = 0; r < RS_R + 2; ++r) xr[r] = __builtin_bit_cast(bf16x8, lds_read16(rb[r] + boff[dx]));
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int p = 0; p < RS_R; ++p)
#pragma unroll
              for (int c = 0; c < RS_CT; ++c)
                acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dy * 3 + dx][c], xr[p + dy],
                                                                   (dx == 0 && dy == 0) ? fzero : acc[p][c], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int p = 0; p < RS_R; ++p)
#pragma unroll
          for (int c = 0; c < RS_CT; ++c) acc[p][c] = fzero;
      }
      tm.stamp(0);
      // this wave's K-quarter of the four output rows -> LDS; the helpers add the quarters up
      char* rbuf = red + parity * RS_RED_B + w * (RS_CT * 1024) + lane * 16;
      parity ^= 1;
#pragma unroll
      for (int j = 0; j < RS_R; ++j)
#pragma unroll
        for (int c = 0; c < RS_CT; ++c)
          *reinterpret_cast<f32x4*>(rbuf + j * (RS_R * RS_CT * 1024) + c * 1024) = acc[j][c];
      tm.stamp(1);
      rs_barrier();
      tm.stamp(2);
    }
  }
}

__global__ __launch_bounds__(RS_NT) void conv3x3_rs_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  char* red = smem + RS_RING_B;                                    // [2][row][k-quarter][ct][lane] f32x4
  float* sc_lds = reinterpret_cast<float*>(red + 2 * RS_RED_B);    // [128]
  float* sh_lds = sc_lds + 128;
  float* stat_red = sh_lds + 128;                                  // [helper][32][2]
  char* rowstage = reinterpret_cast<char*>(stat_red + RS_R * 32 * 2);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  RsTimer tm;
  tm.start(a.dbg != nullptr && blockIdx.x == 0);
  // The BatchNorm fold (four dependent global loads per channel) is done by the compute and finisher waves
  // while the loaders already issue the first row fetches; barrier F publishes scale / shift.
  const bool is_loader = wave >= RS_R && wave < 2 * RS_R;
  if (!is_loader) fd_fold_bn(a, sc_lds, sh_lds, 128, tid < 64 * RS_R ? tid : tid - 64 * RS_R, RS_NT - 64 * RS_R);

  // XCD-aware item order: consecutive workgroup ids go round-robin over the 8 XCDs, so give
  // each XCD a contiguous run of items (neighbouring strips share their halo columns in one L2).
  const int nwg = (int)gridDim.x;
  const int bid = (nwg % 8 == 0) ? ((int)blockIdx.x % 8) * (nwg / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;

  if (wave >= RS_R && (a.dbg_skip & 32)) __builtin_amdgcn_s_setprio(2);
  if (wave >= 2 * RS_R) {
    rs_finisher(a, red, stat_red, rowstage, wave - 2 * RS_R, lane, bid, nwg, tm);
  } else if (wave >= RS_R) {
    const int h = wave - RS_R;
    if (a.pro_mode == 0)
      rs_loader<0>(a, ring, sc_lds, sh_lds, h, lane, bid, nwg, tm);
    else if (a.p_slope == 0.f)
      rs_loader<1>(a, ring, sc_lds, sh_lds, h, lane, bid, nwg, tm);
    else
      rs_loader<2>(a, ring, sc_lds, sh_lds, h, lane, bid, nwg, tm);
  } else {
    rs_compute(a, ring, red, wave, lane, bid, nwg, tm);
  }

  if (tm.on && lane == 0 && wave < 8)
    for (int k = 0; k < 6; ++k) a.dbg[wave * 8 + k] = tm.t[k];
  // ---- common tail: one partial row of statistics per workgroup
  if (a.stats != nullptr) {
    __syncthreads();
    for (int cl = tid; cl < RS_CT * 16; cl += RS_NT) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < RS_R; ++w_) {
        t1 += stat_red[(w_ * RS_CT * 16 + cl) * 2];
        t2 += stat_red[(w_ * RS_CT * 16 + cl) * 2 + 1];
      }
      float* dst = a.stats + ((long long)blockIdx.x * a.stats_cpad + cl) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
    if (a.fin_mean != nullptr) fd_finalize_last_block(a, a.CoutW < RS_CT * 16 ? a.CoutW : RS_CT * 16, tid, ring);
  }
}

int rs_num_cus() {
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

bool conv3x3_rs_fits(const ConvArgs& a, int cout_total) {
  return cout_total <= 32 && a.Cin == 128 && a.pad == 1 && !a.upsample && (a.x_sw % 8) == 0;
}

int conv_dispatch_k3_rs(ConvArgs& a, long long nimg, int cout_total, FdConvInfo* info, long long stats_cap, bool dry,
                        hipStream_t stream) {
  if (!conv3x3_rs_fits(a, cout_total)) FD_FAIL(FD_EUNSUPPORTED, "conv3x3_rs: shape not supported");
  const int ncu = dry ? 256 : rs_num_cus();
  const int strips = (a.Wo + RS_TW - 1) / RS_TW;
  int seg = (a.Ho + RS_R - 1) / RS_R * RS_R;
  auto nitems = [&](int s) { return nimg * strips * ((a.Ho + s - 1) / s); };
  while (nitems(seg) < ncu && seg > 8) seg = (seg / 2 + RS_R - 1) / RS_R * RS_R;   // row segments until every CU has an item
  a.seg_rows = seg;
  a.tiles_x = strips;
  a.tiles_y = (a.Ho + seg - 1) / seg;
  const long long nt = nitems(seg);
  if (nt >= (1ll << 31)) FD_FAIL(FD_EUNSUPPORTED, "conv3x3_rs: too many work items");
  a.ntiles = (int)nt;
  dim3 grid((unsigned)(nt < ncu ? nt : ncu), 1, 1), block(RS_NT, 1, 1);
  a.stats_cpad = RS_CT * 16;
  const unsigned lds = rs_lds_bytes();
  if (info) {
    info->stats_rows = grid.x;
    info->stats_cpad = a.stats_cpad;
    info->grid_x = grid.x;
    info->grid_y = 1;
    info->lds_bytes = lds;
    info->fused_finalize = 1;
  }
  if (dry) return FD_OK;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rs_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv3x3_rs): %s", hipGetErrorString(e));
    attr_done = true;
  }
  if (stats_cap >= 0 && (long long)grid.x * a.stats_cpad * 2 > stats_cap)
    FD_FAIL(FD_EINVAL, "stats workspace too small: need %lld floats, have %lld", (long long)grid.x * a.stats_cpad * 2,
            stats_cap);
  return fd_launch(&conv3x3_rs_kernel, "conv3x3_rs_bn32", grid, block, lds, a, stream);
}
