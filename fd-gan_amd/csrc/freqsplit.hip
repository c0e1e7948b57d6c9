// freqsplit.hip -- the Fusion-discriminator's frequency split (reference: orphaned bytecode
// /root/reference/__pycache__/loss.cpython-36.pyc, semantics in SURVEY Appendix B):
//   LF = Blur:      15x15 isotropic Gaussian (sigma 3), reflection pad 7, same kernel for every
//                   (b, c) plane, optional ImageNet (x - mean) / std first        (loss.py:122-162)
//   HF = Laplacian: depthwise 3x3, ones with centre -8, zero pad 1, NOT normalised (loss.py:205-304)
// Depthwise, HBM-bound kernels on fp32 NCHW planes.  The Gaussian is exactly separable
// (max |k - g g^T| = 7e-18), so Blur is two 15-tap passes over a 46x46 LDS tile; the input
// normalisation is affine and the kernel sums to 1, so it commutes with the blur and is applied
// to the result.  fdgan_fusion_input_nhwc fuses both filters with the layout change the
// discriminator needs: it reads the image once and writes [img, LF(img), HF(img)] as 9 NHWC
// fp16 channels (+ zero padding) -- the tensor D's first 4x4 conv consumes.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int FS_T = 32, FS_R = 7, FS_E = FS_T + 2 * FS_R;   // 32x32 outputs, 46x46 inputs

struct FsArgs {
  const float* x;   // [planes][H][W]
  float* y;         // blur / laplacian output planes (or NULL)
  unsigned short* y_nhwc;   // fused: NHWC fp16 view
  long long yn_sn, yn_sh, yn_sw;
  int H, W, C;      // C = channels per image (plane % C = channel)
  int norm;         // apply (v - mean[c]) / std[c] to the blurred value
  float g[15];      // 1-D Gaussian, sums to 1
  float mean[3], istd[3];
  int tiles_x, tiles_y;
  int mode;         // 0 blur, 1 laplacian, 2 fused NHWC
  int lap_r;        // Laplacian: radius (k - 1) / 2 of the box (0: 1, the 3x3 filter)
  long long y_sn;   // row-streaming form only: elements between the output planes of consecutive images (0: C * H * W, planes packed)
  float* y_copy;    // row-streaming Laplacian only: also receives the INPUT planes, laid out like y (or NULL)
};

__device__ __forceinline__ int reflect(int i, int n) {   // nn.ReflectionPad2d: -i -> i, n-1+i -> n-1-i
  i = i < 0 ? -i : i;
  i = i >= n ? 2 * n - 2 - i : i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);   // tile overhang past the image: value unused, keep it in range
}

__global__ __launch_bounds__(256) void freqsplit_kernel(FsArgs a) {
  __shared__ float tin[FS_E][FS_E + 1];
  __shared__ float tmp[FS_E][FS_T + 1];
  const int plane = blockIdx.y;
  const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
  const int x0 = tx * FS_T, y0 = ty * FS_T;
  const float* xp = a.x + (long long)plane * a.H * a.W;
  const int tid = threadIdx.x;
  // reflected halo tile (the Laplacian's zero padding is applied when it is evaluated)
  for (int i = tid; i < FS_E * FS_E; i += 256) {
    const int r = i / FS_E, c = i - r * FS_E;
    tin[r][c] = xp[(long long)reflect(y0 - FS_R + r, a.H) * a.W + reflect(x0 - FS_R + c, a.W)];
  }
  __syncthreads();
  if (a.mode != 1) {   // horizontal 15-tap pass: 46 rows x 32 columns
    for (int i = tid; i < FS_E * FS_T; i += 256) {
      const int r = i / FS_T, c = i - r * FS_T;
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < 15; ++t) s = fmaf(a.g[t], tin[r][c + t], s);
      tmp[r][c] = s;
    }
  }
  __syncthreads();
  const int ch = plane % a.C, n = plane / a.C;
  for (int i = tid; i < FS_T * FS_T; i += 256) {
    const int r = i / FS_T, c = i - r * FS_T;
    const int oy = y0 + r, ox = x0 + c;
    if (oy >= a.H || ox >= a.W) continue;
    float lf = 0.f, hf = 0.f;
    if (a.mode != 1) {
#pragma unroll
      for (int t = 0; t < 15; ++t) lf = fmaf(a.g[t], tmp[r + t][c], lf);
      if (a.norm) lf = (lf - a.mean[ch]) * a.istd[ch];
    }
    const float ctr = tin[r + FS_R][c + FS_R];
    if (a.mode != 0) {   // k x k box sum with ZERO padding minus k^2 * centre (k = 2 lap_r + 1; 3 x 3 in the network)
      const int lr = a.lap_r > 0 ? a.lap_r : 1;
      float s = 0.f;
      for (int dy = -lr; dy <= lr; ++dy)
        for (int dx = -lr; dx <= lr; ++dx) {
          const int yy = oy + dy, xx = ox + dx;
          s += (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? tin[r + FS_R + dy][c + FS_R + dx] : 0.f;
        }
      hf = s - (float)((2 * lr + 1) * (2 * lr + 1)) * ctr;
    }
    // (mode 2's three 2-byte stores per pixel and plane: a form with all three planes per workgroup, the nine values staged in LDS
    // and ONE 32-byte store per pixel was measured in round 3 -- 39.8 against 35.7 us at 16 x 3 x 256^2, 124 against 140 us at
    // 4 x 3 x 1024^2: the kernel is bound by its 46 x 46 halo tile and scalar LDS reads, not by the stores)
    if (a.mode == 0) {
      a.y[(long long)plane * a.H * a.W + (long long)oy * a.W + ox] = lf;
    } else if (a.mode == 1) {
      a.y[(long long)plane * a.H * a.W + (long long)oy * a.W + ox] = hf;
    } else {

      typedef __attribute__((ext_vector_type(2))) float f2_t;
      unsigned short* q = a.y_nhwc + n * a.yn_sn + oy * a.yn_sh + ox * a.yn_sw;
      const unsigned b01 = fd_pk2<FmtA>((f2_t){ctr, lf});
      const unsigned b2 = fd_pk2<FmtA>((f2_t){hf, 0.f});
      q[ch] = (unsigned short)(b01 & 0xffffu);            // img
      q[a.C + ch] = (unsigned short)(b01 >> 16);          // LF(img)
      q[2 * a.C + ch] = (unsigned short)(b2 & 0xffffu);   // HF(img)
    }
  }
}

// ---- row-streaming form (W % 4 == 0): one WAVE per work item = (plane, 256-column strip, row segment) -----------------
// The tile kernel above reads a 46 x 46 halo for 32 x 32 outputs (2.07x the input) in 4-byte accesses: 1.8 TB/s at
// 1024^2.  Here a wave walks DOWN its strip: per input row every lane loads 16 bytes (4 pixels, coalesced; the two
// 8-pixel halos by four edge lanes, reflected / zeroed there), the row goes through a wave-private LDS line so that a lane
// can read its 20-pixel window (the +-7 horizontal taps cross lanes: this is the "wavefront shuffle", through the LDS
// crossbar as 16-byte reads -- ds_bpermute would need 14 of them per pixel), the horizontal 15-tap (3-tap) sum is formed in
// registers, and the last 15 (3) such rows live in a REGISTER ring -- the loop is unrolled by the ring length so ring
// positions are compile-time -- from which the vertical pass produces one output row per input row.  Column halo: 16 of 272
// pixels (6 %); row halo per segment: 14 rows.  Rows are requested FS_PF ahead so a wave has several loads in flight.
constexpr int FSR_W = 256;


struct FsRowArgs {
  const float* x;
  float* y;
  int H, W, C, norm;
  int strips, segs, seg_rows;
  float g[15];
  float mean[3], istd[3];
  long long y_sn;   // elements between the output planes of consecutive images
  float* y_copy;    // Laplacian: the input planes again, laid out like y (fdgan_fusion_input_nchw), or NULL
};

template <int R, bool REFLECT>   // R = 7: Blur (reflection padding); R = 1: Laplacian box part (zero padding)
__device__ __forceinline__ float fsr_at(const float* plane, int y, int x, int H, int W) {
  if (REFLECT) return plane[(long long)reflect(y, H) * W + reflect(x, W)];
  return (y >= 0 && y < H && x >= 0 && x < W) ? plane[(long long)y * W + x] : 0.f;
}

// Lane crossing by DPP instead of the LDS line (DPP = true): lane i takes the 4-pixel piece of lane i - 1 / i + 1 with ONE
// v_mov_b32_dpp wave_shr:1 / wave_shl:1 per component (a whole-wave shift: gfx9's DPP has it; lane 0 / 63 keep `old`, which
// is the halo piece read out of the edge lanes with v_readlane), and the pieces two lanes away by shifting the shifted value
// once more: 16 DPP moves + 16 readlanes per row against one 16-byte LDS write and five 16-byte LDS reads.
__device__ __forceinline__ float fs_wave_shr1(float old, float v) {   // lane i <- lane i - 1; lane 0 <- old
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float fs_wave_shl1(float old, float v) {   // lane i <- lane i + 1; lane 63 <- old
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float fs_lane(float v, int k) {             // the value lane k holds, in every lane
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}

#ifndef FSR_PF_BLUR
#define FSR_PF_BLUR 3
#endif
template <int R, bool DPP>
__global__ __launch_bounds__(64) void freqsplit_rows_kernel(FsRowArgs a) {
  constexpr bool BLUR = R == 7;
  constexpr int NTAP = 2 * R + 1;
  __shared__ __attribute__((aligned(16))) float line[DPP ? 4 : 8 + FSR_W + 8];
  const int lane = threadIdx.x;
  int item = blockIdx.x;
  const int strip = item % a.strips;
  item /= a.strips;
  const int seg = item % a.segs, plane = item / a.segs;
  const int x0 = strip * FSR_W, r_begin = seg * a.seg_rows, r_end = min(a.H, r_begin + a.seg_rows);
  const float* xp = a.x + (long long)plane * a.H * a.W;
  const int ch = plane % a.C;
  const long long yoff = (long long)(plane / a.C) * a.y_sn + (long long)ch * a.H * a.W;
  float* yp = a.y + yoff;
  const int cx = x0 + 4 * lane;                       // this lane's four columns
  // edge lanes also fetch one 4-pixel piece of the halo: lanes 0, 1 left (x0 - 8, x0 - 4), lanes 2, 3 right (x0 + 256, + 260)
  const int hx = lane < 2 ? x0 - 8 + 4 * lane : x0 + FSR_W + 4 * (lane - 2);
  const int hslot = lane < 2 ? 4 * lane : 8 + FSR_W + 4 * (lane - 2);
  typedef __attribute__((ext_vector_type(4))) float f4;

  // Where a lane's pieces come from, worked out ONCE: W is a multiple of 4 and so is every piece's first column, so a piece lies
  // entirely inside the image or entirely outside.  Outside, Blur reads the reflection -- columns c .. c + 3 are 2 (W - 1) - c down
  // to 2 (W - 1) - c - 3, i.e. the piece that starts at 2 W - 5 - c (left: at -c - 3) in reversed order -- and the Laplacian zeros.
  // Every fetch is then the same two unconditional 16-byte loads: with the edge cases as branches (the first version), hipcc waited
  // for each load at the join right behind it (s_waitcnt vmcnt(0)) and the rows requested ahead were never in flight --
  // 1.4 us per row, the whole memory latency.
  auto piece_src = [&](int c, int& col, bool& rev, bool& zero) __attribute__((always_inline)) {
    const bool out = c < 0 || c >= a.W;
    rev = BLUR && out;
    zero = !BLUR && out;
    col = !out ? c : (c < 0 ? -c - 3 : 2 * a.W - 5 - c);
    col = min(max(col, 0), a.W - 4);        // pieces further out than the window reaches: any readable address
  };
  int vcol, hcol;
  bool vrev, vzero, hrev, hzero;
  piece_src(cx, vcol, vrev, vzero);
  piece_src(lane < 4 ? hx : cx, hcol, hrev, hzero);      // lanes 4 .. 63 fetch their own piece twice (a cache hit) instead of branching
  auto fetch = [&](int iy, f4& v, f4& hv) __attribute__((always_inline)) {   // input row iy (may lie outside the image)
    const int ry = BLUR ? reflect(iy, a.H) : min(max(iy, 0), a.H - 1);
    const float* row = xp + (long long)ry * a.W;
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // a reflected piece starts at a column = 3 mod 4
    v = *reinterpret_cast<const f4u*>(row + vcol), hv = *reinterpret_cast<const f4u*>(row + hcol);
  };
  // ... and what is done to the pieces when they are USED (rows later: touching them in fetch() would wait for the load there)
  auto finish = [&](int iy, f4& v, f4& hv) __attribute__((always_inline)) {
    const f4 t = v, u = hv;
    v = vrev ? f4{t[3], t[2], t[1], t[0]} : t;
    hv = hrev ? f4{u[3], u[2], u[1], u[0]} : u;
    if (!BLUR) {
      const bool rin = iy >= 0 && iy < a.H;
      const bool vz = vzero || !rin, hz = hzero || !rin;
      v = vz ? f4{0.f, 0.f, 0.f, 0.f} : v;
      hv = hz ? f4{0.f, 0.f, 0.f, 0.f} : hv;
    }
  };

  f4 ring[NTAP];       // horizontally filtered rows; ring[k % NTAP] = row (first + k)
  f4 ctr[NTAP];        // Laplacian: the raw centre pixels of the last NTAP rows (the centre row is R rows back)
  constexpr int FSR_PF = BLUR ? FSR_PF_BLUR : (NTAP == 3 ? 3 : NTAP);      // rows requested ahead (divides the unrolled block of NTAP rows)
  f4 pv[FSR_PF], ph[FSR_PF];
  const int first = r_begin - R, last = r_end + R;     // input rows [first, last)
#pragma unroll
  for (int q = 0; q < FSR_PF; ++q) fetch(first + q, pv[q], ph[q]);
  static_assert(NTAP % FSR_PF == 0, "one unrolled block of NTAP rows keeps ring, prefetch and centre positions constant");
  for (int k0 = 0; first + k0 < last; k0 += NTAP) {
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {                   // unrolled: ring / prefetch positions are constants (15 bodies: fits the I-cache; 45 did not)
      const int k = k0 + j, iy = first + k;
      if (iy >= last) continue;          // (a `break` here kept the loop rolled: the ring went to scratch memory)
      f4 v = pv[j % FSR_PF], hv = ph[j % FSR_PF];
      finish(iy, v, hv);
      fetch(iy + FSR_PF, pv[j % FSR_PF], ph[j % FSR_PF]);     // FSR_PF rows ahead (rows past `last` are harmless reads inside the plane / zeros)
      float win[20];                                    // columns cx - 8 .. cx + 11
      if (DPP) {
        // pieces of lanes i - 2, i - 1, i, i + 1, i + 2; past the wave's ends: the halo pieces lanes 0 .. 3 fetched
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float m1 = fs_wave_shr1(fs_lane(hv[e], 1), v[e]);        // lane 0 <- left halo piece x0 - 4 .. x0 - 1
          const float m2 = fs_wave_shr1(fs_lane(hv[e], 0), m1);          // lane 0 <- x0 - 8 .. x0 - 5, lane 1 <- lane 0's m1
          const float p1 = fs_wave_shl1(fs_lane(hv[e], 2), v[e]);        // lane 63 <- right halo piece x0 + 256 .. + 259
          const float p2 = fs_wave_shl1(fs_lane(hv[e], 3), p1);
          win[e] = m2, win[4 + e] = m1, win[8 + e] = v[e], win[12 + e] = p1, win[16 + e] = p2;
        }
      } else {
        // the row through the wave's LDS line (one wave: LDS operations complete in program order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        *reinterpret_cast<f4*>(&line[8 + 4 * lane]) = v;
        if (lane < 4) *reinterpret_cast<f4*>(&line[hslot]) = hv;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          const f4 t = *reinterpret_cast<const f4*>(&line[4 * lane + 4 * q]);
          win[4 * q] = t[0], win[4 * q + 1] = t[1], win[4 * q + 2] = t[2], win[4 * q + 3] = t[3];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      f4 hsum;
      if constexpr (BLUR) {
        // Packed fp32 math wants its two operands in one aligned register pair, and output e reads win[e + 1 + t] -- for a fixed pair
        // of outputs every other tap is misaligned (the first version paid one register move per v_pk_fma_f32 for that).  So the
        // taps are split by parity over OVERLAPPING output pairs: (out0, out1) and (out2, out3) take the odd taps, (out-1, out0),
        // (out1, out2), (out3, out4) the even ones -- every operand is an aligned pair of the window as it was read, 38 packed
        // FMAs and 4 adds per row instead of 60 + 60 moves.
        typedef __attribute__((ext_vector_type(2))) float f2;
        f2 P[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) P[q] = f2{win[2 * q], win[2 * q + 1]};
        f2 qa = {0.f, 0.f}, qb = qa, qc = qa, qd = qa, qz = qa;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          const f2 g2 = {a.g[t], a.g[t]};
          if (t & 1) {
            qa = __builtin_elementwise_fma(g2, P[(1 + t) / 2], qa);
            qc = __builtin_elementwise_fma(g2, P[(3 + t) / 2], qc);
          } else {
            qz = __builtin_elementwise_fma(g2, P[t / 2], qz);
            qb = __builtin_elementwise_fma(g2, P[(2 + t) / 2], qb);
            qd = __builtin_elementwise_fma(g2, P[(4 + t) / 2], qd);
          }
        }
        hsum = f4{qa[0] + qz[1], qa[1] + qb[0], qb[1] + qc[0], qc[1] + qd[0]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s_ = 0.f;
#pragma unroll
          for (int t = 0; t < NTAP; ++t) s_ += win[8 + e + t - R];
          hsum[e] = s_;
        }
      }
      ring[j % NTAP] = hsum;
      if (!BLUR) ctr[j % NTAP] = v;
      const int oy = iy - R;                            // the output row this input row completes
      if (k >= 2 * R && oy >= r_begin && oy < r_end) {
        f4 o = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BLUR) {
          typedef __attribute__((ext_vector_type(2))) float f2;
          f2 o01 = {0.f, 0.f}, o23 = o01;
#pragma unroll
          for (int t = 0; t < NTAP; ++t) {              // rows oy - R .. oy + R = ring positions j - 2R + t
            const f4 hr = ring[(j + 2 * NTAP - 2 * R + t) % NTAP];
            const f2 g2 = {a.g[t], a.g[t]};
            o01 = __builtin_elementwise_fma(g2, f2{hr[0], hr[1]}, o01);
            o23 = __builtin_elementwise_fma(g2, f2{hr[2], hr[3]}, o23);
          }
          o = f4{o01[0], o01[1], o23[0], o23[1]};
        } else {
#pragma unroll
          for (int t = 0; t < NTAP; ++t) {
            const f4 hr = ring[(j + 2 * NTAP - 2 * R + t) % NTAP];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += hr[e];
          }
        }
        if (BLUR) {
          if (a.norm)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (o[e] - a.mean[ch]) * a.istd[ch];
        } else {
          const f4 c = ctr[(j + NTAP - R) % NTAP];   // centre row = the input row R back
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] -= (float)(NTAP * NTAP) * c[e];
          if (a.y_copy != nullptr && cx < a.W) *reinterpret_cast<f4*>(a.y_copy + yoff + (long long)oy * a.W + cx) = c;
        }
        if (cx + 3 < a.W) *reinterpret_cast<f4*>(yp + (long long)oy * a.W + cx) = o;
        else
          for (int e = 0; e < 4; ++e)
            if (cx + e < a.W) yp[(long long)oy * a.W + cx + e] = o[e];
      }
    }
  }
}

// 1-D factor of isotropic_gaussian_kernel(l, sigma), loss.py:153-159, centred in 15 taps (zeros outside the l-tap window: a
// reflection pad of 7 with zero outer taps reads exactly what a reflection pad of l / 2 would; l odd, 1 <= l <= 15)
void gaussian15(float* g, double sigma, int l = 15) {
  double v[15], s = 0.0;
  for (int i = 0; i < 15; ++i) {
    const int d = i - 7;
    v[i] = (d >= -(l / 2) && d <= l / 2) ? exp(-(d * d) / (2.0 * sigma * sigma)) : 0.0;
    s += v[i];
  }
  for (int i = 0; i < 15; ++i) g[i] = (float)(v[i] / s);
}
thread_local int g_blur_l = 15;           // set by the _g entry points around their call into the shared launchers
thread_local double g_blur_sigma = 3.0;

// Adjoint of Blur along ONE axis (the 2-D adjoint is the x pass followed by the y pass; the Gaussian is
// symmetric).  Forward, 1-D: out[i] = sum_t g[t] x[refl(i + t - 7)].  Its transpose is the zero-padded
// correlation z[k] = sum_t g[t] d[k - t + 7] evaluated on the extended range k in [-7, L + 7), with the halo
// folded back onto the samples the reflection read: dx[j] = z[j] + [1 <= j <= 7] z[-j] + [L-8 <= j <= L-2] z[2(L-1)-j].
struct BlurAdjArgs {
  const float* d;
  float* out;
  int H, W, C, axis;      // axis 0: along x (W), 1: along y (H)
  float g[15];
  float scale[3];         // per-channel factor applied on the way out (1/std of the input normalisation, or 1)
  long long total;
};
__global__ __launch_bounds__(256) void blur_adj_kernel(BlurAdjArgs a) {
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  if (u >= a.total) return;
  const int x = (int)(u % a.W);
  long long r = u / a.W;
  const int y = (int)(r % a.H);
  const long long plane = r / a.H;
  const int L = a.axis ? a.H : a.W, j = a.axis ? y : x;
  const long long stride = a.axis ? a.W : 1;
  const float* line = a.d + plane * a.H * a.W + (a.axis ? x : (long long)y * a.W);
  auto z = [&](int k) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 15; ++t) {
      const int i = k - t + 7;
      s = fmaf(a.g[t], (i >= 0 && i < L) ? line[(long long)i * stride] : 0.f, s);
    }
    return s;
  };
  float v = z(j);
  if (j >= 1 && j <= 7) v += z(-j);
  if (j >= L - 8 && j <= L - 2) v += z(2 * (L - 1) - j);
  a.out[u] = v * a.scale[(int)(plane % a.C) % 3];
}

int launch(FsArgs& a, long long planes, hipStream_t stream, const char* name) {
  a.tiles_x = (a.W + FS_T - 1) / FS_T;
  a.tiles_y = (a.H + FS_T - 1) / FS_T;
  FD_REQUIRE(a.H >= 8 && a.W >= 8, "%s: reflection pad 7 needs H, W >= 8", name);
  FD_REQUIRE(planes > 0 && planes < 65536, "%s: plane count %lld", name, planes);
  gaussian15(a.g, g_blur_sigma, g_blur_l);
  const float m[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
  for (int i = 0; i < 3; ++i) {
    a.mean[i] = m[i];
    a.istd[i] = 1.f / sd[i];
  }
  // row-streaming form (one wave per strip segment).  Measured, B=4 @1024^2 / B=16 @256^2: Laplacian 23.8 / 9.2 us against
  // 52.9 / 15.9 us for the tile kernel (4.2 TB/s); Blur 40.5 / 21.6 us against 55.8 / 18.0 us -- its 14 halo rows per
  // 16-row segment cost more than the tile kernel's halo once the planes are small, so small planes stay on the tiles
  const bool rows_pay = a.mode == 1 || a.y_sn != 0 || planes * (long long)a.H * a.W >= (6ll << 20);
  const int lap_r = a.lap_r > 0 ? a.lap_r : 1;
  if (a.mode != 2 && rows_pay && (a.mode == 0 || lap_r <= 3) && a.W % 4 == 0 && a.W >= 16 && a.H >= 16 && ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.y & 15) == 0 &&
      FD_TUNE_GETENV("FDGAN_DEBUG_NO_FS_ROWS") == nullptr) {
    FsRowArgs r{};
    r.x = a.x, r.y = a.y, r.H = a.H, r.W = a.W, r.C = a.C, r.norm = a.norm;
    r.y_sn = a.y_sn ? a.y_sn : (long long)a.C * a.H * a.W, r.y_copy = a.y_copy;
    for (int i = 0; i < 15; ++i) r.g[i] = a.g[i];
    for (int i = 0; i < 3; ++i) r.mean[i] = a.mean[i], r.istd[i] = a.istd[i];
    r.strips = (a.W + FSR_W - 1) / FSR_W;
    // ~16 waves per CU in all; at least 16 (Blur: 14 halo rows are re-filtered per segment) / 8 (Laplacian) rows per segment
    long long segs = 4096 / (planes * r.strips);
    if (segs < 1) segs = 1;
    static const char* sr = FD_TUNE_GETENV("FDGAN_DEBUG_FS_SEG");
    const int min_rows = sr ? atoi(sr) : (a.mode == 1 ? 8 : 16);
    if (segs > a.H / min_rows) segs = a.H / min_rows > 0 ? a.H / min_rows : 1;
    r.seg_rows = (int)((a.H + segs - 1) / segs);
    r.segs = (a.H + r.seg_rows - 1) / r.seg_rows;
    const unsigned grid = (unsigned)(planes * r.strips * r.segs);
    // Lane crossing: measured both ways (tools/freqsplit_bench.py, B=4 @1024^2 / B=16 @256^2, round 3).  Laplacian (3 taps: the
    // crossing IS the kernel): DPP wave shifts 21.6 / 8.4 us (4.66 / 3.0 TB/s) against 24.2 / 10.7 us through the LDS line.
    // Blur (15 taps): 45.9 us with DPP against 41.9 us (round 3).  Round 4: the row fetch without branches (the loads requested
    // ahead really are in flight now) and the horizontal pass on aligned register pairs (106 vector instructions per row, 68 of
    // them v_pk_fma_f32): 40.6 -> 27.1 us at 4 x 3 x 1024^2 = 3.7 TB/s; 3 or 5 rows ahead is the same, 15 is slower (37 us), longer
    // row segments are slower (32 rows: 29.4 us, 64: 37.1 us -- fewer waves), shorter ones too (8: 29.0 us -- more halo rows).
    static const char* dpp_env = FD_TUNE_GETENV("FDGAN_DEBUG_FS_DPP");   // tuning aid: '0' the LDS line, '1' DPP wave shifts
    const bool dpp = dpp_env ? dpp_env[0] == '1' : a.mode == 1;
    if (a.mode == 0) return dpp ? fd_launch(&freqsplit_rows_kernel<7, true>, name, dim3(grid), dim3(64), 0, r, stream)
                                : fd_launch(&freqsplit_rows_kernel<7, false>, name, dim3(grid), dim3(64), 0, r, stream);
    if (lap_r == 2) return dpp ? fd_launch(&freqsplit_rows_kernel<2, true>, name, dim3(grid), dim3(64), 0, r, stream)
                               : fd_launch(&freqsplit_rows_kernel<2, false>, name, dim3(grid), dim3(64), 0, r, stream);
    if (lap_r == 3) return dpp ? fd_launch(&freqsplit_rows_kernel<3, true>, name, dim3(grid), dim3(64), 0, r, stream)
                               : fd_launch(&freqsplit_rows_kernel<3, false>, name, dim3(grid), dim3(64), 0, r, stream);
    return dpp ? fd_launch(&freqsplit_rows_kernel<1, true>, name, dim3(grid), dim3(64), 0, r, stream)
               : fd_launch(&freqsplit_rows_kernel<1, false>, name, dim3(grid), dim3(64), 0, r, stream);
  }
  if (a.y_sn != 0 || a.y_copy != nullptr) FD_FAIL(FD_EUNSUPPORTED, "%s: image-strided output needs the row-streaming form (W %% 4 == 0, W, H >= 16, 16-byte aligned planes)", name);
  return fd_launch(&freqsplit_kernel, name, dim3((unsigned)(a.tiles_x * a.tiles_y), (unsigned)planes), dim3(256), 0, a,
                   stream);
}

}  // namespace

extern "C" int fdgan_blur15_fwd(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w,
                                int use_input_norm, FdStream stream) {
  FD_REQUIRE(x && y, "blur15_fwd: NULL pointer");
  FD_REQUIRE(!use_input_norm || c == 3, "blur15_fwd: use_input_norm needs 3 channels (ImageNet mean/std)");
  FsArgs a{};
  a.x = x;
  a.y = y;
  a.H = (int)h;
  a.W = (int)w;
  a.C = (int)c;
  a.norm = use_input_norm ? 1 : 0;
  a.mode = 0;
  return launch(a, n * c, static_cast<hipStream_t>(stream), "blur15");
}

extern "C" int fdgan_laplacian3_fwd(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w,
                                    FdStream stream) {
  FD_REQUIRE(x && y, "laplacian3_fwd: NULL pointer");
  FsArgs a{};
  a.x = x;
  a.y = y;
  a.H = (int)h;
  a.W = (int)w;
  a.C = (int)c;
  a.mode = 1;
  return launch(a, n * c, static_cast<hipStream_t>(stream), "laplacian3");
}

/* Laplacian(kernel_size).forward for any odd kernel_size <= 15 (loss.py:245-301: ones(k, k) with centre 1 - k^2, zero padding
 * (k - 1) / 2, depthwise): the k x k box sum minus k^2 x the centre.  k = 3, 5, 7 on the row-streaming kernel, larger ones on the tile
 * kernel (its 7-pixel halo is what bounds k).  Self-adjoint like the 3 x 3 one: the backward is the same call on dy. */
extern "C" int fdgan_laplacian_fwd(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, int ksize, FdStream stream) {
  FD_REQUIRE(x && y, "laplacian_fwd: NULL pointer");
  FD_REQUIRE(ksize >= 3 && ksize <= 15 && (ksize & 1), "laplacian_fwd: kernel_size must be odd, 3 .. 15 (got %d)", ksize);
  FsArgs a{};
  a.x = x;
  a.y = y;
  a.H = (int)h;
  a.W = (int)w;
  a.C = (int)c;
  a.mode = 1;
  a.lap_r = ksize / 2;
  return launch(a, n * c, static_cast<hipStream_t>(stream), ksize == 3 ? "laplacian3" : "laplacian_k");
}

/* Laplacian's backward under autograd (loss.py:286-301): the operator is self-adjoint -- a symmetric 3x3 kernel with zero
 * padding -- so dx = Laplacian(dy).  A separate entry point because SURVEY 8(b) names one and a binding reads better with it. */
extern "C" int fdgan_laplacian3_bwd(const float* dy, float* dx, int64_t n, int64_t c, int64_t h, int64_t w, FdStream stream) {
  return fdgan_laplacian3_fwd(dy, dx, n, c, h, w, stream);
}

/* cat([img, Blur(img), Laplacian(img)], 1) as NCHW fp32 planes in ONE buffer (include/fdgan_hip.h). */
extern "C" int fdgan_fusion_input_nchw(const float* img, float* out, int64_t n, int64_t c, int64_t h, int64_t w, int use_input_norm,
                                       FdStream stream) {
  FD_REQUIRE(img && out, "fusion_input_nchw: NULL pointer");
  FD_REQUIRE(!use_input_norm || c == 3, "fusion_input_nchw: use_input_norm needs 3 channels (ImageNet mean/std)");
  FD_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "fusion_input_nchw: empty tensor");
  if (!(w % 4 == 0 && w >= 16 && h >= 16 && ((uintptr_t)img & 15) == 0 && ((uintptr_t)out & 15) == 0 && (h * w) % 4 == 0))
    FD_FAIL(FD_EUNSUPPORTED, "fusion_input_nchw: needs W %% 4 == 0, W, H >= 16 and 16-byte aligned tensors (use the two filters and a concatenation)");
  const long long plane = (long long)h * w;
  FsArgs a{};
  a.x = img, a.H = (int)h, a.W = (int)w, a.C = (int)c;
  a.y_sn = 3 * c * plane;
  a.y = out + c * plane, a.norm = use_input_norm ? 1 : 0, a.mode = 0;
  if (int rc = launch(a, n * c, static_cast<hipStream_t>(stream), "fusion_blur15")) return rc;
  a.y = out + 2 * c * plane, a.norm = 0, a.mode = 1, a.y_copy = out;      // the image itself: channels 0 .. c - 1, same image stride
  return launch(a, n * c, static_cast<hipStream_t>(stream), "fusion_laplacian3_copy");
}

extern "C" int fdgan_fusion_input_nhwc(const float* img, int64_t n, int64_t c, int64_t h, int64_t w,
                                       const FdTensor* y, int use_input_norm, FdStream stream) {
  FD_REQUIRE(img && y && y->ptr, "fusion_input_nhwc: NULL pointer");
  FD_REQUIRE(y->dtype == FD_F16 && y->stride[3] == 1, "fusion_input_nhwc: y must be NHWC fp16");
  FD_REQUIRE(y->n == n && y->h == h && y->w == w && y->c >= 3 * c, "fusion_input_nhwc: y must hold 3*c channels");
  FD_REQUIRE(!use_input_norm || c == 3, "fusion_input_nhwc: use_input_norm needs 3 channels");
  FsArgs a{};
  a.x = img;
  a.y_nhwc = static_cast<unsigned short*>(y->ptr);
  a.yn_sn = y->stride[0];
  a.yn_sh = y->stride[1];
  a.yn_sw = y->stride[2];
  a.H = (int)h;
  a.W = (int)w;
  a.C = (int)c;
  a.norm = use_input_norm ? 1 : 0;
  a.mode = 2;
  return launch(a, n * c, static_cast<hipStream_t>(stream), "fusion_input");
}

/* Blur's backward (loss.py:142-151 under autograd): dx = (1/std) * reflectpad^T(conv^T(dy)); `tmp` is a caller-owned
 * scratch of n*c*h*w floats (the two 1-D adjoint passes). */
extern "C" int fdgan_blur15_bwd(const float* dy, float* tmp, float* dx, int64_t n, int64_t c, int64_t h, int64_t w,
                                int use_input_norm, FdStream stream) {
  FD_REQUIRE(dy && tmp && dx, "blur15_bwd: NULL pointer");
  FD_REQUIRE(h >= 16 && w >= 16, "blur15_bwd: H, W >= 16");
  FD_REQUIRE(!use_input_norm || c == 3, "blur15_bwd: use_input_norm needs 3 channels");
  BlurAdjArgs a{};
  a.H = (int)h, a.W = (int)w, a.C = (int)c;
  gaussian15(a.g, g_blur_sigma, g_blur_l);
  a.total = n * c * h * w;
  const float sd[3] = {0.229f, 0.224f, 0.225f};
  hipStream_t st = static_cast<hipStream_t>(stream);
  a.d = dy, a.out = tmp, a.axis = 0;
  for (int i = 0; i < 3; ++i) a.scale[i] = 1.f;
  int rc = fd_launch(&blur_adj_kernel, "blur15_adj_x", dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, a, st);
  if (rc != FD_OK) return rc;
  a.d = tmp, a.out = dx, a.axis = 1;
  for (int i = 0; i < 3; ++i) a.scale[i] = use_input_norm ? 1.f / sd[i] : 1.f;
  return fd_launch(&blur_adj_kernel, "blur15_adj_y", dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, a, st);
}

/* Blur(l, isotropic_gaussian_kernel(l, sigma)) for any odd l <= 15 and sigma > 0 (loss.py:122-159 builds the module from both;
 * the reference instantiates l = 15, sigma = 3, which is what fdgan_blur15_* are): the same kernels on zero-extended taps. */
extern "C" int fdgan_blur_gauss_fwd(const float* x, float* y, int64_t n, int64_t c, int64_t h, int64_t w, int l, float sigma,
                                    int use_input_norm, FdStream stream) {
  FD_REQUIRE(l >= 1 && l <= 15 && (l & 1) && sigma > 0.f, "blur_gauss_fwd: l = %d (odd, <= 15), sigma = %g", l, (double)sigma);
  g_blur_l = l, g_blur_sigma = sigma;
  const int rc = fdgan_blur15_fwd(x, y, n, c, h, w, use_input_norm, stream);
  g_blur_l = 15, g_blur_sigma = 3.0;
  return rc;
}

extern "C" int fdgan_blur_gauss_bwd(const float* dy, float* tmp, float* dx, int64_t n, int64_t c, int64_t h, int64_t w, int l,
                                    float sigma, int use_input_norm, FdStream stream) {
  FD_REQUIRE(l >= 1 && l <= 15 && (l & 1) && sigma > 0.f, "blur_gauss_bwd: l = %d (odd, <= 15), sigma = %g", l, (double)sigma);
  g_blur_l = l, g_blur_sigma = sigma;
  const int rc = fdgan_blur15_bwd(dy, tmp, dx, n, c, h, w, use_input_norm, stream);
  g_blur_l = 15, g_blur_sigma = 3.0;
  return rc;
}
