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
// The generator's own growth convs (the fast path) now run conv3x3_rs2, at the end of this file, built from
// these timings; this kernel keeps every other shape (ragged strips and rows, fewer filters, bias, epilogue
// activations).
#include <type_traits>

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
  f16x8 wf[9][RS_CT];   // this wave's quarter of the filter: channels [32w, 32w+32), all taps
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < RS_CT; ++c) {
      const bool ok = c < a.ntile_total;
      const u32x4 v = *reinterpret_cast<const u32x4*>(
          a.w + (ok ? ((long long)(w * 9 + t) * a.ntile_total + c) * 512 + lane * 8 : 0));
      wf[t][c] = __builtin_bit_cast(f16x8, ok ? v : zero4);
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
          f16x8 xr[RS_R + 2];
#pragma unroll
          for (int r = 0; r < RS_R + 2; ++r) xr[r] = __builtin_bit_cast(f16x8, lds_read16(rb[r] + boff[dx]));
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int p = 0; p < RS_R; ++p)
#pragma unroll
              for (int c = 0; c < RS_CT; ++c)
                acc[p][c] = fd_mfma_a(wf[dy * 3 + dx][c], xr[p + dy],
                                                                   (dx == 0 && dy == 0) ? fzero : acc[p][c]);
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
      fd_stats_store(a, dst, t1);
      fd_stats_store(a, dst + 1, t2);
    }
    if (a.fin_mean != nullptr) fd_finalize_last_block(a, a.CoutW < RS_CT * 16 ? a.CoutW : RS_CT * 16, tid, ring);
  }
}

// =====================================================================================================================
// conv3x3_rs2 -- the same strips, ring layout and K split, rebuilt around what the phase timers of the kernel above showed
// (128->32 @256^2, cold: loaders alone 62 us, compute + finishers alone 96 us, together 126 us).  It takes the growth convs
// of the generator (32 filters, no bias, no epilogue activation, whole 16-pixel strips and 4-row groups); anything else runs
// the kernel above.
//   * COMPUTE waves: everything static (step t mod 4 fixes the ring rows: immediate offsets, no address arithmetic), the six
//     B reads of a 24-MFMA block are those of the NEXT block -- the third block reads the first column shift of the next
//     step, whose rows were published a barrier ago -- one read per 4 MFMAs, and the partial sums of an output row go to LDS
//     under the next row's MFMAs.  The matrix pipe used to idle for read latency + accumulator drain + 8 stores + barrier
//     skew per iteration (1150 busy cycles of 2500).  With the helpers switched off (FDGAN_DEBUG_PHASES=8) the loop now
//     does the launch's 77 GFLOP in 54-59 us (1.3-1.4 PFLOP/s) at 2400 MHz and 1025 W.
//   * HELPER waves, two per SIMD, each doing what a loader and a finisher did, on every other step: the input row goes
//     global -> registers -> prologue -> ring (no LDS-DMA: 120 cycles of issue per instruction, and the in-place transform
//     read the row back through LDS), everything below the item loop is straight-line code so that hipcc's vmcnt counts
//     stay exact across the register ring, and a wave's two steps are split so that ONLY the five ring writes (plus the
//     fetch and the epilogue of one output row) sit between the barrier that frees the ring rows and the one that publishes
//     them; the prologue arithmetic runs a step earlier, in place, on a row fetched three steps before.
// Measured, 128->32 @256^2 B=16: 92-95 us in the network (3.6 TB/s, 0.45 of the HBM peak, 835 TFLOP/s) against 103-106 us
// for the kernel above; 117-120 us cold against 123-126.  What is left is not an issue limit any more: replayed back to
// back the launch holds the board at its 1400 W limit (rocm-smi: 1380 W, sclk down from 2400 to 2050 MHz;
// tools/power_probe.py), and every restructuring that only removed stall cycles -- 8 waves with 256 registers, deeper fetch
// rings, the transform moved off the critical step -- left the time where it was while the per-step cycle count fell by a
// third.  The remaining levers are joules, not cycles: the LDS traffic (162 KB per step: 72 KB of B fragments, 64 KB of
// K-partials) and the fp32 prologue arithmetic.
// Step t of an item (n iterations, steps 0 .. n): compute = MFMAs of iteration t, its partial sums -> LDS; helper (h, t & 1),
// phase A = ring rows of group t+2 <- its registers, fetch of group t+6 into them, epilogue of output row 4(t-1)+h; in step
// t+1, phase B = store of that output row, prologue on the registers of group t+4.  One s_barrier per step.
constexpr int R2_NT = 768, R2_NR = 16, R2_NH = 8;   // threads, ring rows, helper waves
constexpr int R2_RING_B = R2_NR * RS_ROW_B;   // 73728
__host__ __device__ inline unsigned r2_lds_bytes() {
  return R2_RING_B + 2 * RS_RED_B + 128 * 8 + R2_NH * 32 * 2 * 4 + R2_NH * RowStore<RS_CT>::BYTES;
}
template <int N>
__device__ __forceinline__ void r2_barrier() {   // all LDS operations of this wave but the youngest N are complete
  asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---- compute wave w: input channels [32w, 32w+32).  TM = t mod 4.
template <int TM>
__device__ __forceinline__ void r2_compute_step(const char* const (&xb)[3][2], char* red_lane, const f16x8 (&wf)[9][RS_CT],
                                                f16x8 (&X)[2][RS_R + 2], RsTimer& tm) {
  constexpr int PAR = TM & 1, S0 = (4 * TM) & (R2_NR - 1);
  const f32x4 fzero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[RS_R][RS_CT];
  auto xread = [&](int dx, int r) __attribute__((always_inline)) {   // B fragment of ring row S0 + r, column shift dx
    const int s = (S0 + r) & (R2_NR - 1);
    return __builtin_bit_cast(f16x8, lds_read16(xb[dx][s >> 3] + (s & 7) * RS_ROW_B));
  };
  __builtin_amdgcn_sched_barrier(0);
  // block 0 (dx = 0, fragments X[PAR]) while the dx = 1 fragments arrive
#pragma unroll
  for (int r = 0; r < RS_R + 2; ++r) X[PAR ^ 1][r] = xread(1, r);
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int pp = 0; pp < RS_R; ++pp)
#pragma unroll
      for (int c = 0; c < RS_CT; ++c)
        acc[pp][c] = fd_mfma_a(wf[dy * 3][c], X[PAR][pp + dy], dy == 0 ? fzero : acc[pp][c]);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  // block 1 (dx = 1) while the dx = 2 fragments arrive
#pragma unroll
  for (int r = 0; r < RS_R + 2; ++r) X[PAR][r] = xread(2, r);
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int pp = 0; pp < RS_R; ++pp)
#pragma unroll
      for (int c = 0; c < RS_CT; ++c)
        acc[pp][c] = fd_mfma_a(wf[dy * 3 + 1][c], X[PAR ^ 1][pp + dy], acc[pp][c]);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
  }
  __builtin_amdgcn_sched_barrier(0);
  // block 2 (dx = 2), output row by output row, while the dx = 0 fragments of the next step arrive (garbage after the last
  // step); a finished row's K-quarter goes to LDS under the next row's MFMAs: the helpers add the quarters up
#pragma unroll
  for (int r = 0; r < RS_R + 2; ++r) X[PAR ^ 1][r] = xread(0, RS_R + r);
#pragma unroll
  for (int pp = 0; pp < RS_R; ++pp) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int c = 0; c < RS_CT; ++c)
        acc[pp][c] = fd_mfma_a(wf[dy * 3 + 2][c], X[PAR][pp + dy], acc[pp][c]);
#pragma unroll
    for (int c = 0; c < RS_CT; ++c)
      *reinterpret_cast<f32x4*>(red_lane + PAR * RS_RED_B + pp * (RS_R * RS_CT * 1024) + c * 1024) = acc[pp][c];
  }
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);   // row 0: 6 MFMAs, 2 reads
  __builtin_amdgcn_sched_group_barrier(0x008, 3, 2);
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
  __builtin_amdgcn_sched_group_barrier(0x008, 3, 2);
#pragma unroll
  for (int i = 0; i < 3; ++i) {                        // rows 1 .. 3: 6 MFMAs, the previous row's 2 stores, 4 reads in all
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 2);
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 2);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 2);
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 2);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 2);
  }
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
  __builtin_amdgcn_sched_group_barrier(0x200, 2, 2);
  __builtin_amdgcn_sched_barrier(0);
  tm.stamp(0);
  r2_barrier<0>();
  tm.stamp(2);
}

__device__ __forceinline__ void r2_compute(const ConvArgs& a, const char* ring, char* red, int w, int lane, int bid, int nwg, RsTimer& tm) {
  const int m = lane & 15, g = lane >> 4;
  f16x8 wf[9][RS_CT];   // this wave's quarter of the filter: channels [32w, 32w+32), all taps
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < RS_CT; ++c)
      wf[t][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(a.w + ((long long)(w * 9 + t) * RS_CT + c) * 512 + lane * 8));
  const char* xb[3][2];   // B fragment: pixel m + dx, 16-byte chunk 4w + g, of ring rows 0 / 8
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    xb[dx][0] = ring + (m + dx) * 256 + (((4 * w + g) ^ ((2 * (m + dx)) & 15)) * 16);
    xb[dx][1] = xb[dx][0] + 8 * RS_ROW_B;
  }
  char* red_lane = red + w * (RS_CT * 1024) + lane * 16;
  f16x8 X[2][RS_R + 2];
  r2_barrier<0>();   // F

  for (int item = bid; item < a.ntiles; item += nwg) {
    const int n = rs_item(a, item).n_iter;
    r2_barrier<0>();   // P: groups 0 and 1 are in the ring
#pragma unroll
    for (int r = 0; r < RS_R + 2; ++r) X[0][r] = __builtin_bit_cast(f16x8, lds_read16(xb[0][0] + r * RS_ROW_B));
    int t = 0;
    for (; t + 4 <= n; t += 4) {
      r2_compute_step<0>(xb, red_lane, wf, X, tm);
      r2_compute_step<1>(xb, red_lane, wf, X, tm);
      r2_compute_step<2>(xb, red_lane, wf, X, tm);
      r2_compute_step<3>(xb, red_lane, wf, X, tm);
    }
    if (t < n) r2_compute_step<0>(xb, red_lane, wf, X, tm);
    if (t + 1 < n) r2_compute_step<1>(xb, red_lane, wf, X, tm);
    if (t + 2 < n) r2_compute_step<2>(xb, red_lane, wf, X, tm);
    r2_barrier<0>();   // step n: the helpers finish the last row group
  }
}

// ---- helper wave (h, par): input row h of the groups = par (mod 2), output row h of the iterations != par (mod 2).
// Everything below the item loop is straight-line code (selects instead of branches, harmless duplicate or L2-hot fetches
// and ring writes instead of skipped ones): the compiler's s_waitcnt vmcnt counts stay exact across the register ring only
// without control flow between a fetch and its use -- with uniform branches around the fetches every wait degraded to
// vmcnt(0), which serialised each step with the memory latency of the row fetched IN it.
template <int XMODE>
__device__ __forceinline__ void r2_helper(const ConvArgs& a, char* ring, const char* red, const float* sc_lds, const float* sh_lds,
                                          float* stat_red, char* rowstage, int hw, int lane, int bid, int nwg, RsTimer& tm) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  const int h = hw & 3, par = hw >> 2;
  const int tc = lane & 15, tp = lane >> 4;   // fetch / prologue: 16-byte channel chunk, pixel within a group of 4
  const int m = lane & 15, g = lane >> 4;     // epilogue (MFMA result layout): pixel m, couts g*4.. of tile c
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, h0 = s0, h1 = s0;
  bool first = true;
  // unit `it` of a lane = (pixel it*4 + lane/16, channel chunk lane%16) of the 18-pixel ring row; unit 4 exists for pixels
  // 16, 17 only: lanes 32..63 repeat lanes 0..31 (same address, same value, same ring slot)
  // byte offset inside a ring row: [18 px][16 slots], slot = chunk ^ (2 px & 15); units 2, 3 = units 0, 1 + 8 pixels (same slot)
  auto xf_of = [&](int it) __attribute__((always_inline)) {
    const int p = it * 4 + (it == 4 ? (tp & 1) : tp);
    return p * 256 + ((tc ^ ((2 * p) & 15)) * 16);
  };
  const int xf0 = xf_of(0), xf1 = xf_of(1), xf4 = xf_of(4);
  auto xf_off = [&](int it) __attribute__((always_inline)) { return it == 4 ? xf4 : ((it & 1) ? xf1 : xf0) + (it >> 1) * 2048; };
  const unsigned row_pitch_b = (unsigned)a.x_sh * 2u;
  f32x2_t st1[RS_CT][2], st2[RS_CT][2];   // (no bias: the dispatcher sends biased convs to the kernel above)
#pragma unroll
  for (int c = 0; c < RS_CT; ++c)
#pragma unroll
    for (int r = 0; r < 2; ++r) st1[c][r] = st2[c][r] = (f32x2_t){0.f, 0.f};
  char* tb = rowstage + hw * RowStore<RS_CT>::BYTES;
  char* tb_w = tb + m * RowStore<RS_CT>::PITCH + g * 8;                              // staging: MFMA layout in ...
  const char* tb_r = tb + (lane >> 2) * RowStore<RS_CT>::PITCH + (lane & 3) * 16;   // ... one pixel quarter per lane out
  const unsigned y_lane = (unsigned)((lane >> 2) * a.y_sw + (lane & 3) * 8);
  const char* red_lane = red + h * (RS_R * RS_CT * 1024) + lane * 16;

  for (int item = bid; item < a.ntiles; item += nwg) {
    const RsItem I = rs_item(a, item);
    const int n = I.n_iter;
    if (a.dbg_skip & 8) {   // measurement aid: the compute waves alone
      if (first) r2_barrier<0>();
      first = false;
      for (int t = 0; t < n + 2; ++t) r2_barrier<0>();
      continue;
    }
    const int x0 = I.sx * RS_TW - 1, y0 = I.sg * a.seg_rows - 1;   // image coords of ring pixel 0 / ring row 0
    const char* img = reinterpret_cast<const char*>(a.x + (long long)I.n * a.x_sn);
    unsigned src_off[5], keepbits = 0;   // columns outside the image: fetched from the clamped column, zeroed after the prologue
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int px = x0 + it * 4 + (it == 4 ? (tp & 1) : tp);
      src_off[it] = (unsigned)(min(max(px, 0), a.Ws - 1) * a.x_sw + tc * 8) * 2u;
      keepbits |= (px >= 0 && px < a.Ws) ? (1u << it) : 0u;
    }
    auto keep = [&](int it) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_sbfe((int)keepbits, it, 1); };
    // first row this wave stores: iteration 0 for par = 1 (step 1), iteration 1 for par = 0 (step 2); then every other one
    unsigned short* yrow = reinterpret_cast<unsigned short*>(a.y) + (long long)I.n * a.y_sn + (long long)(I.sx * RS_TW) * a.y_sw +
                           (long long)(I.sg * a.seg_rows + h + (par ? 0 : RS_R)) * a.y_sh;

    // Two register sets.  In its k-th cycle (steps t = par + 2k and t+1) the wave
    //   A (step t):    writes S[k % 2] -- group t+2, transformed a step ago -- to the ring, refills it with the fetch of group
    //                  t+6, and finishes output row 4(t-1)+h (partials -> sum, statistics -> staging tile);
    //   B (step t+1):  stores the staged row and applies the prologue IN PLACE to S[(k+1) % 2], group t+4, fetched three steps ago.
    // Only the five ring writes sit between the barrier that frees the ring rows and the one that publishes them: with the
    // transform in the same step the helpers' step was 3000 cycles against 1600 of the compute waves, however many of them
    // shared the work.  A fetch has three steps to arrive.
    u32x4 S[2][5], tmp[5];
    auto issue_row = [&](int q, bool wanted, u32x4 (&dst)[5]) __attribute__((always_inline)) {
      // ring row q of the item; `wanted` false (a group past the item's last): the item's first row again (L2-hot, never used)
      const int gy = min(max(y0 + (wanted ? q : 0), 0), a.Hs - 1);
      const char* rowp = img + (unsigned long long)gy * row_pitch_b;
#pragma unroll
      for (int it = 0; it < 5; ++it) dst[it] = *reinterpret_cast<const u32x4*>(rowp + src_off[it]);
    };
    auto transform_row = [&](int q, const u32x4 (&src)[5], u32x4 (&dst)[5]) __attribute__((always_inline)) {
      const unsigned rowmask = (y0 + q >= 0 && y0 + q < a.Hs) ? 0xffffffffu : 0u;   // zero padding is post-activation
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        u32x4 v = XMODE == 0 ? src[it] : fd_xform8_r(src[it], s0, s1, h0, h1, XMODE == 1 ? 0.f : a.p_slope);
        dst[it] = v & (keep(it) & rowmask);
      }
    };
    auto write_row = [&](int q, const u32x4 (&src)[5]) __attribute__((always_inline)) {
      char* row = ring + (q & (R2_NR - 1)) * RS_ROW_B;
#pragma unroll
      for (int it = 0; it < 5; ++it) lds_write16(row + xf_off(it), src[it]);
    };
    const char* rbuf = red_lane;
    // sum of the four K-partials, statistics; the row goes to the wave's staging tile as fp16 [pixel][32 channels] and is read
    // back one pixel quarter per lane and stored in phase B
    auto epilogue = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < RS_CT; ++c) {
        f32x4 part[RS_R];
#pragma unroll
        for (int k = 0; k < RS_R; ++k) part[k] = *reinterpret_cast<const f32x4*>(rbuf + (k * RS_CT + c) * 1024);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          f32x2_t t = ((f32x2_t){part[0][2 * r], part[0][2 * r + 1]} + (f32x2_t){part[1][2 * r], part[1][2 * r + 1]}) +
                      ((f32x2_t){part[2][2 * r], part[2][2 * r + 1]} + (f32x2_t){part[3][2 * r], part[3][2 * r + 1]});
          st1[c][r] += t;
          st2[c][r] = __builtin_elementwise_fma(t, t, st2[c][r]);
          v[2 * r] = t[0];
          v[2 * r + 1] = t[1];
        }
        *reinterpret_cast<u32x2*>(tb_w + c * 32) = fd_pk4<FmtA>(v);
      }
    };
    auto flush_staged = [&]() __attribute__((always_inline)) {
      *reinterpret_cast<u32x4*>(yrow + y_lane) = *reinterpret_cast<const u32x4*>(tb_r);
      yrow += (long long)(2 * RS_R) * a.y_sh;
    };

    // Before the first step: groups 0 (par 0: rows h and, for h < 2, 4 + h) and 1 (par 1: row 6 + h) into the ring, this wave's
    // first group (2 + par) transformed in S[0], its second (4 + par) in flight in S[1]
    const int g0 = 2 + par;
    issue_row(par ? 6 + h : h, true, S[0]);
    issue_row(par ? 6 + h : (h < 2 ? 4 + h : h), true, S[1]);   // (where there is no second row: the first once more, same ring row)
    issue_row(4 * g0 + 2 + h, g0 < n, tmp);
    if (first) {   // barrier F: the compute waves have folded BatchNorm into (scale, shift) meanwhile
      first = false;
      r2_barrier<0>();
      s0 = *reinterpret_cast<const f32x4*>(sc_lds + tc * 8);
      s1 = *reinterpret_cast<const f32x4*>(sc_lds + tc * 8 + 4);
      h0 = *reinterpret_cast<const f32x4*>(sh_lds + tc * 8);
      h1 = *reinterpret_cast<const f32x4*>(sh_lds + tc * 8 + 4);
    }
    transform_row(par ? 6 + h : h, S[0], S[0]);
    write_row(par ? 6 + h : h, S[0]);
    transform_row(par ? 6 + h : (h < 2 ? 4 + h : h), S[1], S[1]);
    write_row(par ? 6 + h : (h < 2 ? 4 + h : h), S[1]);
    issue_row(4 * (g0 + 2) + 2 + h, g0 + 2 < n, S[1]);
    transform_row(4 * g0 + 2 + h, tmp, S[0]);
    r2_barrier<0>();   // P

    // phase A of the cycle starting at step t (set `cur`), phase B (set `nxt`)
    auto phase_a = [&](int t, u32x4 (&cur)[5], auto epi) __attribute__((always_inline)) {
      write_row(4 * (t + 2) + 2 + h, cur);
      tm.stamp(4);   // (waits for the five writes to COMPLETE: with the transform pinned into phase B this is pure LDS queueing behind the
                     // compute waves' fragment reads -- 51 of 170 k ticks: the kernel's limit is LDS traffic, 162 KB per step)
      const int qf = 4 * (t + 6) + 2 + h;
      const char* rowp = img + (unsigned long long)min(max(y0 + (t + 6 < n ? qf : 0), 0), a.Hs - 1) * row_pitch_b;
#pragma unroll
      for (int it = 0; it < 5; ++it) cur[it] = *reinterpret_cast<const u32x4*>(rowp + src_off[it]);
      tm.stamp(5);
      if (decltype(epi)::value) {
        rbuf = red_lane + ((t - 1) & 1) * RS_RED_B;
        epilogue();
      }
      tm.stamp(0);
      r2_barrier<0>();
      tm.stamp(1);
    };
    auto phase_b = [&](int t, u32x4 (&nxt)[5], auto epi) __attribute__((always_inline)) {   // step t + 1
      if (decltype(epi)::value) flush_staged();
      transform_row(4 * (t + 4) + 2 + h, nxt, nxt);
      // The transformed row must EXIST before this step's barrier: the barrier asm only clobbers memory, so hipcc was free to sink
      // the whole transform (and the vmcnt wait for the row it works on) below it -- into the next step, in front of the five
      // ring writes, the one place every other wave of the workgroup waits for (round 4, phase timers: 84 of the 96 k ticks of
      // phase A sat before the first ring write).  An empty asm that reads and writes the five registers pins them here.
      asm volatile("" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]), "+v"(nxt[4]));
      tm.stamp(2);
      r2_barrier<0>();
      tm.stamp(3);
    };
    const std::integral_constant<bool, true> YES;
    const std::integral_constant<bool, false> NO;
    // steps 0 .. n; cycles at t = par, par + 2, ...: the first cycle of par 0 (t = 0) has nothing to finish
    int t = par;
    if (par) r2_barrier<0>();   // step 0 belongs to the other parity
    if (par == 0) {
      phase_a(0, S[0], NO);
      if (1 <= n) phase_b(0, S[1], NO);
    } else if (1 <= n) {
      phase_a(1, S[0], YES);
      if (2 <= n)
        phase_b(1, S[1], YES);
      else
        flush_staged();
    }
    t += 2;
    for (; t + 3 <= n; t += 4) {   // two cycles: sets S[1], S[0]
      phase_a(t, S[1], YES);
      phase_b(t, S[0], YES);
      phase_a(t + 2, S[0], YES);
      phase_b(t + 2, S[1], YES);
    }
    if (t <= n) {
      phase_a(t, S[1], YES);
      if (t + 1 <= n) {
        phase_b(t, S[0], YES);
        if (t + 2 <= n) {
          phase_a(t + 2, S[0], YES);
          if (t + 3 <= n)
            phase_b(t + 2, S[1], YES);
          else
            flush_staged();
        }
      } else {
        flush_staged();
      }
    }
  }
  if (a.stats != nullptr) {
#pragma unroll
    for (int c = 0; c < RS_CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1v = fd_row_sum16(st1[c][r >> 1][r & 1]), s2v = fd_row_sum16(st2[c][r >> 1][r & 1]);
        if (m == 0) {
          const int idx = (hw * RS_CT * 16 + c * 16 + g * 4 + r) * 2;
          stat_red[idx] = s1v;
          stat_red[idx + 1] = s2v;
        }
      }
  }
}

__global__ __launch_bounds__(R2_NT) void conv3x3_rs2_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  char* red = smem + R2_RING_B;                                    // [2][row][k-quarter][ct][lane] f32x4
  float* sc_lds = reinterpret_cast<float*>(red + 2 * RS_RED_B);    // [128]
  float* sh_lds = sc_lds + 128;
  float* stat_red = sh_lds + 128;                                  // [helper][32][2]
  char* rowstage = reinterpret_cast<char*>(stat_red + R2_NH * 32 * 2);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The BatchNorm fold (four dependent global loads per channel) is done by the compute waves while the helpers already
  // issue the first row fetches; barrier F publishes scale / shift.
  const int nwg = (int)gridDim.x;
  const int bid = (nwg % 8 == 0) ? ((int)blockIdx.x % 8) * (nwg / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;   // XCD-aware, as above
  RsTimer tm;   // measurement aid (tools/conv_bench.py FDGAN_TIMING=1)
  tm.start(a.dbg != nullptr && blockIdx.x == 0);
  if (wave < RS_R) {
    fd_fold_bn(a, sc_lds, sh_lds, 128, tid, 64 * RS_R);
    r2_compute(a, ring, red, wave, lane, bid, nwg, tm);
  } else {
    const int hw = wave - RS_R;
    if (a.pro_mode == 0)
      r2_helper<0>(a, ring, red, sc_lds, sh_lds, stat_red, rowstage, hw, lane, bid, nwg, tm);
    else if (a.p_slope == 0.f)
      r2_helper<1>(a, ring, red, sc_lds, sh_lds, stat_red, rowstage, hw, lane, bid, nwg, tm);
    else
      r2_helper<2>(a, ring, red, sc_lds, sh_lds, stat_red, rowstage, hw, lane, bid, nwg, tm);
  }
  if (tm.on && lane == 0 && wave < 8)
    for (int k = 0; k < 6; ++k) a.dbg[wave * 8 + k] = tm.t[k];
  // ---- common tail: one partial row of statistics per workgroup
  if (a.stats != nullptr) {
    __syncthreads();
    for (int cl = tid; cl < RS_CT * 16; cl += R2_NT) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < R2_NH; ++w_) {
        t1 += stat_red[(w_ * RS_CT * 16 + cl) * 2];
        t2 += stat_red[(w_ * RS_CT * 16 + cl) * 2 + 1];
      }
      float* dst = a.stats + ((long long)blockIdx.x * a.stats_cpad + cl) * 2;
      fd_stats_store(a, dst, t1);
      fd_stats_store(a, dst + 1, t2);
    }
    if (a.fin_mean != nullptr) fd_finalize_last_block(a, RS_CT * 16, tid, ring);
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
  const int ncu = fd_cus(dry ? 256 : rs_num_cus());
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
  // the second-generation kernel: 32 stored channels, full-width strips, whole row groups, no bias, no epilogue activation
  bool gen2 = a.CoutW == RS_CT * 16 && a.ntile_total == RS_CT && a.Cout >= RS_CT * 16 && a.y_vec16 && a.Wo % RS_TW == 0 && a.Ho % RS_R == 0 &&
              seg % RS_R == 0 && a.e_slope == 1.f && a.Ws == a.Wo && a.Hs == a.Ho && a.bias == nullptr;
  if (const char* e = FD_TUNE_GETENV("FDGAN_DEBUG_RS2")) gen2 = gen2 && e[0] != '0';
  const unsigned lds = gen2 ? r2_lds_bytes() : rs_lds_bytes();
  if (gen2) block = dim3(R2_NT, 1, 1);
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
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rs2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv3x3_rs): %s", hipGetErrorString(e));
    attr_done = true;
  }
  if (stats_cap >= 0 && (long long)grid.x * a.stats_cpad * 2 > stats_cap)
    FD_FAIL(FD_EINVAL, "stats workspace too small: need %lld floats, have %lld", (long long)grid.x * a.stats_cpad * 2,
            stats_cap);
  if (gen2) return fd_launch(&conv3x3_rs2_kernel, "conv3x3_rs2_bn32", grid, block, lds, a, stream);
  return fd_launch(&conv3x3_rs_kernel, "conv3x3_rs_bn32", grid, block, lds, a, stream);
}
