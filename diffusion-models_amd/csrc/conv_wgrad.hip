// Weight gradient of the implicit-GEMM convolution on bf16 MFMA (gfx950).
//
//   dw[n][tap][c] += sum_m dy[m][n] * X(m, tap)[c]
//
// The reduction index m (output pixel) is the SLOW index of both operands in memory (NHWC), so both MFMA
// operands are column slices of row-major [pixel][channel] LDS tiles. The tiles are filled by
// buffer_load ... lds (zero rows for padding / rows beyond M) and the fragments are read with
// ds_read_b64_tr_b16, the gfx950 transposing LDS read: two reads give a lane its 8 consecutive pixels of one
// channel. 32-byte units of each row are XOR-swizzled (source side) so the 8 rows one half-wave touches per
// transposing read land on different banks.
//
// One workgroup = one (tap, 128|64 output channels, 128|64 input channels) tile over one slice of the pixel
// axis (split-K); partial tiles are transposed through LDS and added with 256-byte-contiguous fp32 atomics
// into dw, which therefore ACCUMULATES (callers zero it or keep the running gradient in it).
#include "dm_common.h"

namespace {

constexpr int BKM = 64;  // pixels per step
constexpr int NTHREADS = 256;

struct WgradArgs {
  const void* x1;
  const void* x2;
  const void* dy;
  float* dw;
  int C1, C2, Cin, Cout;
  int B, Hs, Ws, Hi, Wi, Ho, Wo, KH, KW, stride, pad, ups;
  int M, splitk, rows_per_split, n_co_tiles, n_ci_tiles, xcd_major;
  unsigned x1_bytes, x2_bytes, dy_bytes;
  unsigned magic_hw, shift_hw, magic_w, shift_w;
};

__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned magic, unsigned shift) {
  return (__umulhi(n, magic) + n) >> shift;
}

// swizzle of the 16-byte piece index within a row, by row; PPR = pieces per row (16: 256-byte rows, 8: 128-byte)
template <int PPR>
__device__ __forceinline__ int swz(int row) {
  if (PPR == 16) return (((row & 3) | ((row >> 1) & 4)) << 1);
  return ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1);
}

template <int TW>  // tile width in channels (64 or 128)
struct Tile {
  static constexpr int PPR = TW / 8;
  static constexpr int ROW_BYTES = TW * 2;
  static constexpr int BYTES = BKM * ROW_BYTES;
  static constexpr int PIECES = BKM * PPR / NTHREADS;  // per thread
  static constexpr int ROW_STEP = NTHREADS / PPR;
};

template <int TW>
__device__ __forceinline__ bf16x8_t read_frag(const char* tile, int colbase, int kk, int lane) {
  using T = Tile<TW>;
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int piece = (colbase >> 3) + (p >> 1);
  const int inner = (p & 1) * 8;
  const int r0 = kk * 32 + 8 * g + q;
  const int r1 = r0 + 4;
  const char* p0 = tile + r0 * T::ROW_BYTES + ((piece ^ swz<T::PPR>(r0)) << 4) + inner;
  const char* p1 = tile + r1 * T::ROW_BYTES + ((piece ^ swz<T::PPR>(r1)) << 4) + inner;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
  s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int TCO, int TCI>
__global__ __launch_bounds__(NTHREADS, 2) void conv_wgrad_kernel(WgradArgs a) {
  using TY = Tile<TCO>;
  using TX = Tile<TCI>;
  constexpr int STAGE = TY::BYTES + TX::BYTES;
  constexpr int EP_PITCH = TCI + 4;
  constexpr int EP_BYTES = TCO * EP_PITCH * 4;
  constexpr int TOTAL = (2 * STAGE > EP_BYTES) ? 2 * STAGE : EP_BYTES;
  constexpr int MI = TCO / 32, NJ = TCI / 32;  // 16x16 tiles per wave (waves 2x2)
  __shared__ __attribute__((aligned(16))) char smem[TOTAL];

  int ks, inner;
  const int per = a.KH * a.KW * a.n_co_tiles * a.n_ci_tiles;
  if (a.xcd_major) {
    // slices of the pixel axis are dealt to XCDs: all taps/tiles of one slice re-read the same rows from one L2
    dm_xcd_tile(blockIdx.x, per, ks, inner);
    if (ks >= a.splitk) return;
  } else {
    ks = blockIdx.x / per;
    inner = blockIdx.x - ks * per;
  }
  const int tiles = a.n_co_tiles * a.n_ci_tiles;
  const int tap = inner / tiles;
  const int rest = inner - tap * tiles;
  const int tco = rest / a.n_ci_tiles, tci = rest - tco * a.n_ci_tiles;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int co0 = tco * TCO, ci0 = tci * TCI;
  const bool second = ci0 >= a.C1;
  const int csrc = second ? a.C2 : a.C1;
  const int cioff = second ? ci0 - a.C1 : ci0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(second ? a.x2 : a.x1), 0,
                                                                 second ? a.x2_bytes : a.x1_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);

  const int m_begin = ks * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);
  const int nsteps = (m_end - m_begin + BKM - 1) / BKM;

  // staging geometry
  const int yrow = tid / TY::PPR, yslot = tid % TY::PPR;
  const int ypiece = yslot ^ swz<TY::PPR>(yrow);
  const int xrow = tid / TX::PPR, xslot = tid % TX::PPR;
  const int xpiece = xslot ^ swz<TX::PPR>(xrow);
  // output channels beyond Cout (partial tile) read zeros
  const bool ycol_ok = co0 + ypiece * 8 < a.Cout;
  const bool xcol_ok = cioff + xpiece * 8 < csrc;

  auto stage = [&](int step, int buf) {
    char* sy = smem + buf * STAGE;
    char* sx = sy + TY::BYTES;
    const int mb = m_begin + step * BKM;
#pragma unroll
    for (int i = 0; i < TY::PIECES; ++i) {
      int m = mb + yrow + TY::ROW_STEP * i;
      unsigned off = (m < m_end && ycol_ok) ? (unsigned)((m * a.Cout + co0 + ypiece * 8) * 2) : DM_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, DM_LDS(sy + (wave * 64 + NTHREADS * i) * 16), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TX::PIECES; ++i) {
      int m = mb + xrow + TX::ROW_STEP * i;
      unsigned off = DM_OOB;
      if (m < m_end && xcol_ok) {
        unsigned b = fastdiv((unsigned)m, a.magic_hw, a.shift_hw);
        unsigned r = (unsigned)m - b * (unsigned)(a.Ho * a.Wo);
        unsigned oh = fastdiv(r, a.magic_w, a.shift_w);
        unsigned ow = r - oh * (unsigned)a.Wo;
        int ih = (int)oh * a.stride - a.pad + kh, iw = (int)ow * a.stride - a.pad + kw;
        if (ih >= 0 && iw >= 0 && ih < a.Hi && iw < a.Wi) {
          ih >>= a.ups; iw >>= a.ups;
          off = (unsigned)(((((int)b * a.Hs + ih) * a.Ws + iw) * csrc + cioff + xpiece * 8) * 2);
        }
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, DM_LDS(sx + (wave * 64 + NTHREADS * i) * 16), 16, off, 0, 0, 0);
    }
  };

  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) stage(0, 0);
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nsteps) stage(s + 1, (s + 1) & 1);
    const char* sy = smem + (s & 1) * STAGE;
    const char* sx = sy + TY::BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[MI], bfr[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = read_frag<TCO>(sy, wm * (TCO / 2) + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = read_frag<TCI>(sx, wn * (TCI / 2) + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: transpose through LDS, then row-contiguous fp32 atomics
  __syncthreads();
  float* ep = (float*)smem;
  const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ep[(wm * (TCO / 2) + i * 16 + fq * 4 + r) * EP_PITCH + wn * (TCI / 2) + j * 16 + frow] = acc[i][j][r];
  __syncthreads();
  const int T = a.KH * a.KW;
  for (int item = tid; item < TCO * TCI; item += NTHREADS) {
    int row = item / TCI, col = item - row * TCI;
    int co = co0 + row, ci = ci0 + col;
    if (co < a.Cout)
      atomicAdd(a.dw + ((size_t)co * T + tap) * a.Cin + ci, ep[row * EP_PITCH + col]);
  }
}

void make_magic(unsigned d, unsigned& magic, unsigned& shift) {
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  magic = (unsigned)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  shift = s;
}

}  // namespace

extern "C" int dm_conv2d_wgrad(const dm_conv_desc* d, const void* x1, const void* x2, const void* dy, float* dw,
                               void* stream) {
  DM_REQUIRE(d && x1 && dy && dw, DM_ERR_ARG, "dm_conv2d_wgrad: null argument");
  DM_REQUIRE(d->C1 > 0 && d->C1 % 64 == 0 && d->C2 >= 0 && d->C2 % 64 == 0, DM_ERR_SHAPE,
             "dm_conv2d_wgrad: input channels (%d, %d) must be multiples of 64", d->C1, d->C2);
  DM_REQUIRE((d->C2 == 0) == (x2 == nullptr), DM_ERR_ARG, "dm_conv2d_wgrad: x2 and C2 disagree");
  DM_REQUIRE(d->Cout > 0 && d->Cout % 8 == 0, DM_ERR_SHAPE, "dm_conv2d_wgrad: Cout=%d must be a multiple of 8", d->Cout);
  DM_REQUIRE(!d->transposed, DM_ERR_UNSUPPORTED, "dm_conv2d_wgrad: transposed descriptor");
  DM_REQUIRE(d->stride == 1 || d->stride == 2, DM_ERR_UNSUPPORTED, "dm_conv2d_wgrad: stride %d", d->stride);
  DM_REQUIRE(!(d->ups && d->stride != 1), DM_ERR_UNSUPPORTED, "dm_conv2d_wgrad: fused upsample needs stride 1");
  DM_REQUIRE(((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)dy | (uintptr_t)dw) % 16 == 0, DM_ERR_ARG, "dm_conv2d_wgrad: pointers must be 16-byte aligned");
  const int Hi = d->Hs << d->ups, Wi = d->Ws << d->ups;
  DM_REQUIRE((Hi + 2 * d->pad - d->KH) / d->stride + 1 == d->Ho && (Wi + 2 * d->pad - d->KW) / d->stride + 1 == d->Wo,
             DM_ERR_SHAPE, "dm_conv2d_wgrad: in %dx%d k%d s%d p%d does not give out %dx%d", Hi, Wi, d->KH, d->stride, d->pad, d->Ho, d->Wo);
  const long npix_src = (long)d->B * d->Hs * d->Ws;
  const long M = (long)d->B * d->Ho * d->Wo;
  DM_REQUIRE(npix_src * d->C1 * 2 < 0x7fffffffL && npix_src * d->C2 * 2 < 0x7fffffffL && M * d->Cout * 2 < 0x7fffffffL,
             DM_ERR_UNSUPPORTED, "dm_conv2d_wgrad: tensor larger than 2 GiB");

  WgradArgs a;
  a.x1 = x1; a.x2 = x2; a.dy = dy; a.dw = dw;
  a.C1 = d->C1; a.C2 = d->C2; a.Cin = d->C1 + d->C2; a.Cout = d->Cout;
  a.B = d->B; a.Hs = d->Hs; a.Ws = d->Ws; a.Hi = Hi; a.Wi = Wi; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.ups = d->ups;
  a.M = (int)M;
  a.x1_bytes = (unsigned)(npix_src * d->C1 * 2);
  a.x2_bytes = (unsigned)(npix_src * d->C2 * 2);
  a.dy_bytes = (unsigned)(M * d->Cout * 2);
  make_magic((unsigned)(d->Ho * d->Wo), a.magic_hw, a.shift_hw);
  make_magic((unsigned)d->Wo, a.magic_w, a.shift_w);

  const bool wide_co = d->Cout % 128 == 0;
  const bool wide_ci = d->C1 % 128 == 0 && d->C2 % 128 == 0;
  const int tco = wide_co ? 128 : 64, tci = wide_ci ? 128 : 64;
  a.n_co_tiles = dm_cdiv(d->Cout, tco);
  a.n_ci_tiles = a.Cin / tci;
  const int per = d->KH * d->KW * a.n_co_tiles * a.n_ci_tiles;
  // split the pixel axis so the launch has >= ~2 workgroups per CU, but keep >= 8 steps per workgroup
  const int steps_total = dm_cdiv(M, BKM);
  int splitk = dm_cdiv(768, per);
  const int max_split = steps_total / 8 > 0 ? steps_total / 8 : 1;
  if (splitk > max_split) splitk = max_split;
  if (splitk < 1) splitk = 1;
  a.rows_per_split = dm_cdiv(steps_total, splitk) * BKM;
  a.splitk = dm_cdiv(M, a.rows_per_split);
  a.xcd_major = a.splitk >= 8;
  const int ks_pad = a.xcd_major ? (a.splitk + 7) / 8 * 8 : a.splitk;
  dim3 grid(ks_pad * per), block(NTHREADS);
  hipStream_t st = (hipStream_t)stream;
  if (wide_co && wide_ci) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), grid, block, 0, st, a);
  else if (wide_co) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64>), grid, block, 0, st, a);
  else if (wide_ci) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), grid, block, 0, st, a);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
