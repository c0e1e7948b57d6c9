// conv1x1_bwd.hip -- data gradient of the dense-layer bottleneck (1x1, Cout_fwd = Cy <= 128 filters, any Cin_fwd = C <= 1024)
// fused with the backward of its BatchNorm + ReLU prologue, as a streaming kernel.
//
//   da[p][c]  = sum_k dy[p][k] * W[k][c]                  (k over the Cy forward filters)
//   v         = da * act'(bn(x[p][c]));   sums (v, v * x) per channel;   G[p][c] += gamma*rstd * v   (or T = v)
//
// The generic implicit-GEMM kernel (conv_igemm.h, MK = 1) moves 2.7-2.9 TB/s on these shapes: a 128-pixel x 128-channel
// tile per workgroup with four K steps is all prologue and epilogue, and every channel tile re-reads dy.  Here the
// work is organised around the data that is big -- x and G, 3 C values per pixel against Cy of dy:
//   * persistent workgroups (2 per CU, 4 waves) walk 64-pixel tiles of the flattened N*H*W axis; a tile's dy rows
//     (64 x Cy) are staged once in LDS and serve every channel tile;
//   * per 128-channel tile the filter fragments (32 KB, L2-resident) are staged through registers; the x and G rows
//     of a (pixel tile, channel tile) pair are requested one whole pair AHEAD (two register sets used alternately), so
//     HBM reads are in flight while the previous pair's epilogue computes and stores -- with the requests issued in the
//     same iteration every phase serialised (570 us for a 256x256 layer; each phase alone ran at HBM speed);
//   * the epilogue is the row phase of the MK kernels: the accumulator tile goes through a wave-private LDS
//     transposition, every lane owns 8 channels of 4 pixels, whole 16-byte pieces of pixel rows in and out;
//   * BatchNorm's two sums stay in LDS across the workgroup's tiles (fixed summation order) and leave as ONE partial
//     row per workgroup.
// Reference: autograd of conv1 / norm1 / relu1 of torchvision's _DenseLayer as used by
// /root/reference/models/dehaze1113.py:713-724.
#include <stdlib.h>

#include "conv_igemm.h"

namespace {

constexpr int B1_PX = 64;                        // pixels per tile (one 16-pixel MFMA column block per wave)
constexpr int B1_CT = 128;                       // channels per channel tile
constexpr int B1_TBP = B1_CT * 2 + 16;           // staging pitch of one pixel (272 B)
constexpr int B1_TB = 16 * B1_TBP;               // per-wave transposition area
// LDS-only barrier: hipcc's __syncthreads() also waits for every outstanding global load, which would drain the x / G
// rows requested ahead of the MFMAs
#define B1_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct Bwd1Args {
  const unsigned short* dy;   // [P][dy_pitch] bf16, Cy channels used
  int dy_pitch, Cy;
  const unsigned short* w;    // chunk32 filter image of the (flipped = same for 1x1) forward filter: A fragments
  int ntile_total;            // ceil(C / 16)
  const unsigned short* x;    // forward input, [P][x_pitch]
  int x_pitch;
  unsigned short* g;          // gradient buffer (accumulate) or dpre
  int g_pitch;
  int C, nct;                 // channels, channel tiles
  long long P;
  int ntiles;                 // P / 64
  int mode, acc;              // 1 activation only, 2 BatchNorm + activation; acc 1: G += gamma*rstd*v, 2: G = gamma*rstd*v, 0: G = v
  float slope, eps;
  const float *mean, *var, *gamma, *beta;
  float* partial;             // [gridDim.x][nct * 128][2] or NULL
  int dbg;                    // FDGAN_DEBUG_PHASES (results wrong): 1 no filter loads, 2 no MFMAs, 4 no row phase, 8 no x / G loads, 16 no dy tile
};

__global__ __launch_bounds__(256, 2) void conv1x1_bwd_kernel(Bwd1Args a) {
  extern __shared__ __attribute__((aligned(16))) char b1_lds[];
  const int kch = a.Cy / 32;                                  // k chunks (<= 4)
  char* dyt = b1_lds;                                         // [64 px][Cy * 2 B], 16-byte columns XOR-swizzled by the pixel
  char* wt = dyt + B1_PX * a.Cy * 2;                          // [kch][8 tiles][1 KB] A fragments of the channel tile
  char* tb0 = wt + kch * 8 * 1024;                            // 4 x transposition areas
  float* red = reinterpret_cast<float*>(tb0 + 4 * B1_TB);     // [4 waves][128][2]
  float* stats = red + 4 * 128 * 2;                           // [nct * 128][2] running sums of this workgroup
  // scale / shift of every channel, computed once (up to 5 channel tiles: two workgroups still fit a CU), else of the
  // current channel tile only
  const bool sc_once = a.nct <= 5;
  float* scall = stats + a.nct * 256;                         // [cpad] scale, [cpad] shift  |  [128] scale, [128] shift
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kgl = lane >> 4;
  char* tb = tb0 + wave * B1_TB;
  const int cpad = a.nct * B1_CT;
  for (int i = tid; i < a.nct * 256; i += 256) stats[i] = 0.f;
  for (int c = tid; sc_once && c < cpad; c += 256) {
    float sc = 1.f, sh = 0.f;
    if (a.mode == 2 && c < a.C) {
      const float gm = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
      sc = gm / sqrtf(a.var[c] + a.eps);
      sh = bt - a.mean[c] * sc;
    }
    scall[c] = sc;
    scall[cpad + c] = sh;
  }
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int dyrow = a.Cy * 2;                                 // bytes per staged dy pixel
  const int dcols = a.Cy / 8;                                 // 16-byte columns per dy pixel: 4, 8 or 16
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  typedef __attribute__((ext_vector_type(4))) float f4_t;
  // row phase: lanes of a quad own the same 8-channel piece of 4 consecutive pixels, so the per-channel sums reduce
  // with two DPP quad swaps
  const int piece = lane >> 2, q0 = lane & 3;
  const int my_tiles = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int n_it = my_tiles * a.nct;                          // (tile, channel tile) pairs of this workgroup, in order

  // x / G rows of one (tile, channel tile) pair: requested one whole iteration ahead, so that HBM reads are in flight
  // while the previous pair's row phase computes and stores (two register sets, used alternately)
  auto request = [&](int it, u32x4 (&xv)[4], u32x4 (&gv)[4]) __attribute__((always_inline)) {
    const int tl = it / a.nct, ct = it - tl * a.nct;
    const long long p0 = (long long)((int)blockIdx.x + tl * (int)gridDim.x) * B1_PX;
    const int cg = ct * B1_CT + piece * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long p = p0 + wave * 16 + i * 4 + q0;
      xv[i] = gv[i] = zero4;
      if (it < n_it && cg < a.C && !(a.dbg & 8)) {
        xv[i] = *reinterpret_cast<const u32x4*>(a.x + p * a.x_pitch + cg);
        if (a.acc == 1) gv[i] = *reinterpret_cast<const u32x4*>(a.g + p * a.g_pitch + cg);
      }
    }
  };
  auto step = [&](int it, u32x4 (&xv)[4], u32x4 (&gv)[4], u32x4 (&xn)[4], u32x4 (&gn)[4]) __attribute__((always_inline)) {
    const int tl = it / a.nct, ct = it - tl * a.nct;
    const long long p0 = (long long)((int)blockIdx.x + tl * (int)gridDim.x) * B1_PX;
    const int c0 = ct * B1_CT;
    // ---- filter fragments of this channel tile -> registers
    u32x4 wr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * 256;                            // 16-byte unit of the [kch][8][64 lanes] fragment block
      const int kc = f >> 9, t8 = (f >> 6) & 7, ln = f & 63;
      const int tile16 = ct * 8 + t8;
      wr[i] = zero4;
      if (kc < kch && tile16 < a.ntile_total && !(a.dbg & 1)) wr[i] = *reinterpret_cast<const u32x4*>(a.w + ((long long)kc * a.ntile_total + tile16) * 512 + ln * 8);
    }
    u32x4 dyr[4];                                             // this tile's dy rows (first channel tile only)
    if (ct == 0 && !(a.dbg & 16))
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = tid + i * 256;
        dyr[i] = zero4;
        if (u < B1_PX * dcols) dyr[i] = *reinterpret_cast<const u32x4*>(a.dy + (p0 + u / dcols) * a.dy_pitch + (u % dcols) * 8);
      }
    float sc_t = 1.f, sh_t = 0.f;                             // per-tile mode: this channel tile's coefficients (threads 0-127)
    if (!sc_once && tid < 128 && a.mode == 2 && c0 + tid < a.C) {
      const int c = c0 + tid;
      const float gm = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
      sc_t = gm / sqrtf(a.var[c] + a.eps);
      sh_t = bt - a.mean[c] * sc_t;
    }
    B1_BARRIER();                                             // previous pair: fragments read, red written, coefficients used
    if (!sc_once && tid < 128) {
      scall[tid] = sc_t;
      scall[128 + tid] = sh_t;
    }
    if (it > 0 && a.partial != nullptr) {                     // previous pair's sums: fixed order over the four waves
      const int pc0 = ((it - 1) % a.nct) * B1_CT, c = tid >> 1, which = tid & 1;
      const float t = (red[(0 * 128 + c) * 2 + which] + red[(1 * 128 + c) * 2 + which]) +
                      (red[(2 * 128 + c) * 2 + which] + red[(3 * 128 + c) * 2 + which]);
      stats[(pc0 + c) * 2 + which] += t;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + i * 256;
      if ((f >> 9) < kch) lds_write16(wt + f * 16, wr[i]);
    }
    if (ct == 0 && !(a.dbg & 16))   // column c16 of pixel q lands at column c16 ^ (q mod columns): conflict-free B-fragment reads
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = tid + i * 256, q = u / dcols, c16 = u % dcols;
        if (u < B1_PX * dcols) lds_write16(dyt + q * dyrow + ((c16 ^ (q & (dcols - 1))) << 4), dyr[i]);
      }
    B1_BARRIER();
    // ---- MFMA: wave = 16 pixels x 128 channels
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = wave * 16 + m;
    if (!(a.dbg & 2))
      for (int kc = 0; kc < kch; ++kc) {
        const int c16 = kc * 4 + kgl;
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, lds_read16(dyt + q * dyrow + ((c16 ^ (q & (dcols - 1))) << 4)));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bf16x8 afr = __builtin_bit_cast(bf16x8, lds_read16(wt + ((kc * 8 + j) * 64 + lane) * 16));
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, acc[j], 0, 0, 0);
        }
      }
    request(it + 1, xn, gn);                                  // next pair's rows: in flight during this row phase
    // ---- row phase: transpose through the wave's staging area, mask, sums, G += gamma*rstd*v (or store v)
    if (a.dbg & 4) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x2 bits = __builtin_bit_cast(u32x2, __builtin_convertvector((f4_t){acc[j][0], acc[j][1], acc[j][2], acc[j][3]}, bf16x4_t));
      *reinterpret_cast<u32x2*>(tb + m * B1_TBP + j * 32 + kgl * 8) = bits;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int cg = c0 + piece * 8;
    const bool ch_ok = cg < a.C;
    float sc8[8], sh8[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc8[e] = sc_once ? scall[cg + e] : scall[piece * 8 + e];
      sh8[e] = sc_once ? scall[cpad + cg + e] : scall[128 + piece * 8 + e];
      s1[e] = s2[e] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ql = i * 4 + q0;
      const f32x8 da = __builtin_convertvector(__builtin_bit_cast(bf16x8, lds_read16(tb + ql * B1_TBP + piece * 16)), f32x8);
      const f32x8 fx = __builtin_convertvector(__builtin_bit_cast(bf16x8, xv[i]), f32x8);
      f32x8 o = __builtin_convertvector(__builtin_bit_cast(bf16x8, gv[i]), f32x8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pre = fmaf(fx[e], sc8[e], sh8[e]);
        const float v = (cg + e < a.C) ? da[e] * (pre > 0.f ? 1.f : a.slope) : 0.f;
        s1[e] += v;
        s2[e] += v * fx[e];
        o[e] = a.acc ? fmaf(sc8[e], v, o[e]) : v;
      }
      if (ch_ok) *reinterpret_cast<u32x4*>(a.g + (p0 + wave * 16 + ql) * a.g_pitch + cg) = __builtin_bit_cast(u32x4, __builtin_convertvector(o, bf16x8));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (a.partial != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += __shfl_xor(s1[e], 1, 64);
        s2[e] += __shfl_xor(s2[e], 1, 64);
        s1[e] += __shfl_xor(s1[e], 2, 64);
        s2[e] += __shfl_xor(s2[e], 2, 64);
      }
      if (q0 == 0)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[(wave * 128 + piece * 8 + e) * 2] = s1[e];
          red[(wave * 128 + piece * 8 + e) * 2 + 1] = s2[e];
        }
    }
  };

  u32x4 xa[4], ga[4], xb[4], gb[4];
  B1_BARRIER();                                               // scall / stats initialised
  request(0, xa, ga);
  for (int it = 0; it < n_it; it += 2) {
    step(it, xa, ga, xb, gb);
    if (it + 1 < n_it) step(it + 1, xb, gb, xa, ga);
  }
  if (a.partial != nullptr) {
    B1_BARRIER();
    if (n_it > 0) {
      const int pc0 = ((n_it - 1) % a.nct) * B1_CT, c = tid >> 1, which = tid & 1;
      const float t = (red[(0 * 128 + c) * 2 + which] + red[(1 * 128 + c) * 2 + which]) +
                      (red[(2 * 128 + c) * 2 + which] + red[(3 * 128 + c) * 2 + which]);
      stats[(pc0 + c) * 2 + which] += t;
    }
    B1_BARRIER();
    float* dst = a.partial + (long long)blockIdx.x * a.nct * 256;
    for (int i = tid; i < a.nct * 256; i += 256) dst[i] = stats[i];
  }
}

}  // namespace

bool conv1x1_bwd_fits(const FdTensor* dy, const FdTensor* fwd_x, const FdTensor* dpre) {
  auto dense = [](const FdTensor* t) {
    return t->stride[3] == 1 && t->stride[1] == t->w * t->stride[2] && t->stride[0] == t->h * t->stride[1] && t->stride[2] % 8 == 0 &&
           ((uintptr_t)t->ptr & 15) == 0;
  };
  const long long P = dpre->n * dpre->h * dpre->w;
  return (dy->c == 32 || dy->c == 64 || dy->c == 128) && dense(dy) && dense(fwd_x) && dense(dpre) && P % B1_PX == 0 &&
         dpre->c <= 1024 && dpre->stride[2] >= (dpre->c + 7) / 8 * 8 && fwd_x->stride[2] >= (dpre->c + 7) / 8 * 8 &&
         P * fwd_x->stride[2] < (1ll << 40) && getenv("FDGAN_DEBUG_NO_BWD1X1S") == nullptr;
}

/* rows_out / cpad_out: shape of the partial-sum block written when `partial` is given. */
int conv1x1_bwd_launch(const FdTensor* dy, const void* w_packed, const FdTensor* fwd_x, const FdPrologue* pro, const FdTensor* dpre,
                       int accumulate, float* partial, long long capacity_floats, long long* rows_out, long long* cpad_out,
                       hipStream_t stream) {
  Bwd1Args a{};
  a.dy = static_cast<const unsigned short*>(dy->ptr), a.dy_pitch = (int)dy->stride[2], a.Cy = (int)dy->c;
  a.w = static_cast<const unsigned short*>(w_packed);
  a.C = (int)dpre->c, a.ntile_total = (a.C + 15) / 16, a.nct = (a.C + B1_CT - 1) / B1_CT;
  a.x = static_cast<const unsigned short*>(fwd_x->ptr), a.x_pitch = (int)fwd_x->stride[2];
  a.g = static_cast<unsigned short*>(dpre->ptr), a.g_pitch = (int)dpre->stride[2];
  a.P = dpre->n * dpre->h * dpre->w, a.ntiles = (int)(a.P / B1_PX);
  const bool norm = pro && pro->mean;
  const int act = pro ? pro->act : FD_ACT_NONE;
  a.mode = norm ? 2 : 1, a.acc = accumulate;
  a.slope = act == FD_ACT_RELU ? 0.f : (act == FD_ACT_LEAKY02 ? 0.2f : 1.f);
  if (norm) a.mean = pro->mean, a.var = pro->var, a.gamma = pro->gamma, a.beta = pro->beta, a.eps = pro->eps;
  long long grid = a.ntiles < 512 ? a.ntiles : 512;            // two resident workgroups per CU
  a.partial = norm ? partial : nullptr;
  if (norm) FD_REQUIRE(partial && grid * a.nct * 256 <= capacity_floats, "conv2d_bwd_data: workspace too small (%lld floats needed)", grid * a.nct * 256);
  static const char* ph = getenv("FDGAN_DEBUG_PHASES");
  a.dbg = ph ? atoi(ph) : 0;
  if (rows_out) *rows_out = grid;
  if (cpad_out) *cpad_out = a.nct * 128;
  const unsigned lds = (unsigned)(B1_PX * a.Cy * 2 + (a.Cy / 32) * 8 * 1024 + 4 * B1_TB + 4 * 128 * 2 * 4 + a.nct * 256 * 4 +
                                  (a.nct <= 5 ? a.nct : 1) * 256 * 4);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv1x1_bwd): %s", hipGetErrorString(e));
    attr_done = true;
  }
  return fd_launch(&conv1x1_bwd_kernel, "conv1x1_bwd_stream", dim3((unsigned)grid), dim3(256), lds, a, stream);
}
