// Implicit-GEMM convolution on bf16 MFMA (gfx950): forward and data-gradient.
//
//   y[m][n] = sum_{tap, c} X(m, tap)[c] * W[n][tap][c]  (+ bias[n]) (+ res[m][n])
//
// m = (b, oh, ow) output pixel, n = output channel, activations NHWC bf16, weights [N][T][C] bf16 ("OHWI").
// X(m, tap) is gathered per row on the fly:
//   regular     ih = oh*stride - pad + kh                      (optionally >>1: fused nearest x2 upsample)
//   transposed  ih = (oh + pad - kh) / stride if divisible      (data gradient of a strided conv)
// and rows that fall into the padding come back as zeros from the buffer descriptor's range check, so the
// main loop has no branches. The channel axis may be the concatenation of two tensors (x1 | x2).
//
// Tile: 128 pixels x BN channels x 64 reduction elements per step, 4 waves (2x2), mfma_f32_16x16x32_bf16.
// Operand tiles are written straight into LDS by buffer_load ... lds (16 B per lane, lane-linear image), with
// the 16-byte chunks of each 128-byte row XOR-swizzled on the SOURCE side (chunk ^ (row & 7)) so that the
// ds_read_b128 fragment reads are bank-conflict free. Two LDS buffers: the loads of step s+1 are in flight
// while step s is multiplied. The epilogue transposes the accumulators through LDS so that bias/residual are
// read and the output written as whole 16-byte, row-contiguous pieces.
#include "dm_common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int NTHREADS = 256;

struct ConvArgs {
  const void* x1;
  const void* x2;
  const void* w;
  const float* bias;
  const void* res;
  void* y;
  int C1, C2, Cin, Cout;
  int B, Hs, Ws, Hi, Wi, Ho, Wo;
  int KH, KW, stride, pad, ups, transposed;
  int M, n_mtiles, n_ntiles;
  unsigned x1_bytes, x2_bytes, w_bytes;
  // Data gradient of a 3x3 / stride-2 / pad-1 convolution onto an even-sized image: an output pixel only meets the
  // taps whose parity matches its own (1, 2, 2 or 4 of the 9), so the rows of a tile are taken from ONE parity class
  // (tile mt -> class mt / tpc, rows in class-major order) and the tile runs that class's taps only.
  int subpix, tpc, Mq;
};

template <int BN>
struct Smem {
  static constexpr int A_BYTES = BM * BK * 2;
  static constexpr int B_BYTES = BN * BK * 2;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int EP_PITCH = BN + 4;  // floats
  static constexpr int EP_BYTES = BM * EP_PITCH * 4;
  static constexpr int TOTAL = (2 * STAGE > EP_BYTES) ? 2 * STAGE : EP_BYTES;
};

template <int BN>
__global__ __launch_bounds__(NTHREADS, 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int WN = BN / 2;      // columns per wave
  constexpr int NJ = WN / 16;     // 16-wide MFMA tiles per wave along N
  constexpr int MI = 4;           // 64 rows per wave
  constexpr int B_PIECES = BN / 32;  // 16-byte pieces of the weight tile per thread
  using S = Smem<BN>;
  __shared__ __attribute__((aligned(16))) char smem[S::TOTAL];

  int mt, nt;
  dm_xcd_tile(blockIdx.x, a.n_ntiles, mt, nt);
  if (mt >= a.n_mtiles) return;
  const int n0 = nt * BN;
  const int cls = a.subpix ? mt / a.tpc : 0;
  const int m0 = (a.subpix ? mt - cls * a.tpc : mt) * BM;     // first row within the class (or of the whole tensor)
  const int py = cls >> 1, px = cls & 1;
  // row of the tile -> output pixel index (-1: none)
  auto row_pixel = [&](int row) -> int {
    const int m = m0 + row;
    if (!a.subpix) return m < a.M ? m : -1;
    if (m >= a.Mq) return -1;
    const int hq = a.Ho >> 1, wq = a.Wo >> 1;
    const int b = m / (hq * wq), r = m - b * (hq * wq);
    const int yq = r / wq, xq = r - yq * wq;
    return (b * a.Ho + 2 * yq + py) * a.Wo + 2 * xq + px;
  };
  // the taps a tile runs: all of them, or those of its parity class (kernel row kh meets output row oh iff
  // oh + pad - kh is even: kh = 1 for even rows, kh = 0 and 2 for odd ones; columns alike)
  const int kh_begin = a.subpix ? (py ? 0 : 1) : 0, kw_begin = a.subpix ? (px ? 0 : 1) : 0;
  const int k_step = a.subpix ? 2 : 1;
  const int n_taps = a.subpix ? (py ? 2 : 1) * (px ? 2 : 1) : a.KH * a.KW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x1), 0, a.x1_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x2 ? a.x2 : a.x1), 0,
                                                                 a.x2 ? a.x2_bytes : 0u, 0x00020000);
  __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);

  // ---- per-thread staging geometry: 4 rows of the pixel tile, fixed 16-byte chunk
  const int srow = tid >> 3;                       // + 32*i
  const int chunk = (tid & 7) ^ (srow & 7);        // global chunk that lands in LDS slot (tid & 7)
  int rb[4], rh[4], rw_[4];                        // image index * Hs, oh/ow pre-offset; rb < 0: row beyond M
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = row_pixel(srow + 32 * i);
    if (m >= 0) {
      int hw = a.Ho * a.Wo;
      int b = m / hw, r = m - b * hw;
      int oh = r / a.Wo, ow = r - oh * a.Wo;
      rb[i] = b * a.Hs;
      rh[i] = a.transposed ? (oh + a.pad) : (oh * a.stride - a.pad);
      rw_[i] = a.transposed ? (ow + a.pad) : (ow * a.stride - a.pad);
    } else {
      rb[i] = -1; rh[i] = 0; rw_[i] = 0;
    }
  }
  unsigned wrow[B_PIECES];
#pragma unroll
  for (int i = 0; i < B_PIECES; ++i) {
    int n = n0 + srow + 32 * i;
    wrow[i] = (n < a.Cout) ? (unsigned)(n * (a.KH * a.KW) * a.Cin * 2 + chunk * 16) : DM_OOB;
  }

  // staging cursor (tap, channel offset) and the source pixel of each of the 4 rows for that tap
  int s_kh = kh_begin, s_kw = kw_begin, s_cc = 0;
  int pix[4];
  auto set_tap = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ih, iw;
      bool ok = rb[i] >= 0;
      if (a.transposed) {
        int th = rh[i] - s_kh, tw = rw_[i] - s_kw;
        ok = ok && th >= 0 && tw >= 0 && ((th | tw) & (a.stride - 1)) == 0;
        ih = th >> (a.stride - 1);  // stride is 1 or 2
        iw = tw >> (a.stride - 1);
        ok = ok && ih < a.Hi && iw < a.Wi;
      } else {
        ih = rh[i] + s_kh; iw = rw_[i] + s_kw;
        ok = ok && ih >= 0 && iw >= 0 && ih < a.Hi && iw < a.Wi;
        ih >>= a.ups; iw >>= a.ups;
      }
      pix[i] = ok ? ((rb[i] + ih) * a.Ws + iw) : -1;
    }
  };
  set_tap();

  auto stage = [&](int buf) {
    char* sa = smem + buf * S::STAGE;
    char* sb = sa + S::A_BYTES;
    const bool second = s_cc >= a.C1;
    const int csrc = second ? a.C2 : a.C1;
    const int coff = (second ? s_cc - a.C1 : s_cc) * 2 + chunk * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned off = pix[i] >= 0 ? (unsigned)(pix[i] * csrc * 2 + coff) : DM_OOB;
      char* dst = sa + (wave * 64 + 256 * i) * 16;
      if (second)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, DM_LDS(dst), 16, off, 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, DM_LDS(dst), 16, off, 0, 0, 0);
    }
    const unsigned koff = (unsigned)(((s_kh * a.KW + s_kw) * a.Cin + s_cc) * 2);
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
      unsigned off = wrow[i] == DM_OOB ? DM_OOB : wrow[i] + koff;
      char* dst = sb + (wave * 64 + 256 * i) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, DM_LDS(dst), 16, off, 0, 0, 0);
    }
    // advance the cursor
    s_cc += BK;
    if (s_cc >= a.Cin) {
      s_cc = 0;
      s_kw += k_step;
      if (s_kw >= a.KW) { s_kw = kw_begin; s_kh += k_step; }
      set_tap();
    }
  };

  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes within a stage buffer), k-step kk adds 4 chunks
  const int frow = lane & 15, fq = lane >> 4;
  int a_off[MI][2], b_off[NJ][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int row = wm * 64 + i * 16 + frow;
      a_off[i][kk] = row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int row = wn * WN + j * 16 + frow;
      b_off[j][kk] = S::A_BYTES + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4);
    }
  }

  const int nsteps = n_taps * (a.Cin / BK);
  stage(0);
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nsteps) stage((s + 1) & 1);
    const char* sbuf = smem + (s & 1) * S::STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[MI], bfr[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8_t*)(sbuf + a_off[i][kk]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = *(const bf16x8_t*)(sbuf + b_off[j][kk]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: accumulators -> LDS (fp32, padded rows) -> bias / residual / bf16 -> 16-byte stores
  __syncthreads();
  float* ep = (float*)smem;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ep[(wm * 64 + i * 16 + fq * 4 + r) * S::EP_PITCH + wn * WN + j * 16 + frow] = acc[i][j][r];
  __syncthreads();
  constexpr int CPR = BN / 8;  // 16-byte output pieces per row
  bf16_t* y = (bf16_t*)a.y;
  const bf16_t* res = (const bf16_t*)a.res;
#pragma unroll 2
  for (int it = 0; it < (BM * CPR) / NTHREADS; ++it) {
    int item = tid + NTHREADS * it;
    int row = item / CPR, cch = item - row * CPR;
    const int m = row_pixel(row), n = n0 + cch * 8;
    if (m >= 0 && n < a.Cout) {
      const float* src = ep + row * S::EP_PITCH + cch * 8;
      f32x4_t v0 = *(const f32x4_t*)src, v1 = *(const f32x4_t*)(src + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (a.bias) {
        f32x4_t b0 = *(const f32x4_t*)(a.bias + n), b1 = *(const f32x4_t*)(a.bias + n + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
      }
      size_t o = (size_t)m * a.Cout + n;
      if (res) {
        bf16x8_t rv = *(const bf16x8_t*)(res + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
      }
      *(bf16x8_t*)(y + o) = dm_pack8(v);
    }
  }
}

}  // namespace

int dm_conv_patch_try(const dm_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                      const void* res, void* y, hipStream_t st, const dm_gn_desc* g, const float* gamma,
                      const float* beta, const float* ss, const float* pre, void* y_norm, float* mean,
                      float* rstd, const dm_conv_proj* proj);
int dm_conv1x1_try(const dm_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                   const void* res, void* y, hipStream_t st);
int dm_conv_patch_gn_ok(const dm_conv_desc* d, const dm_gn_desc* g);
int dm_conv_patch_proj_ok(const dm_conv_desc* d, const dm_conv_proj* p);
extern int g_dm_use_patch_kernel;
extern int g_dm_conv_gn_fuse;
extern int g_dm_dgrad_subpixel;
extern int g_dm_conv_proj_fuse;
extern int g_dm_timing_skip;
extern int g_dm_timing_twice;

static int conv2d_igemm(const dm_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                        const void* res, void* y, void* stream, const dm_gn_desc* g, const float* gamma,
                        const float* beta, const float* ss, const float* pre, void* y_norm, float* mean,
                        float* rstd, const dm_conv_proj* proj = nullptr);

extern "C" int dm_conv2d_proj_fusable(const dm_conv_desc* d, const dm_conv_proj* p) {
  return d && p && g_dm_use_patch_kernel && g_dm_conv_proj_fuse && dm_conv_patch_proj_ok(d, p);
}

extern "C" int dm_conv2d_igemm_proj(const dm_conv_desc* d, const void* x1, const void* x2, const void* w,
                                    const float* bias, const dm_conv_proj* p, void* y, void* stream) {
  DM_REQUIRE(p && p->x1 && p->w, DM_ERR_ARG, "dm_conv2d_igemm_proj: null argument");
  DM_REQUIRE((p->C2 == 0) == (p->x2 == nullptr), DM_ERR_ARG, "dm_conv2d_igemm_proj: projection x2 and C2 disagree");
  DM_REQUIRE(((uintptr_t)p->x1 | (uintptr_t)p->x2 | (uintptr_t)p->w | (uintptr_t)p->bias) % 16 == 0, DM_ERR_ARG,
             "dm_conv2d_igemm_proj: pointers must be 16-byte aligned");
  DM_REQUIRE(dm_conv2d_proj_fusable(d, p), DM_ERR_UNSUPPORTED,
             "dm_conv2d_igemm_proj: this layer does not take the fused projection (ask dm_conv2d_proj_fusable first)");
  return conv2d_igemm(d, x1, x2, w, bias, nullptr, y, stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                      nullptr, p);
}

extern "C" int dm_conv2d_igemm(const dm_conv_desc* d, const void* x1, const void* x2, const void* w,
                               const float* bias, const void* res, void* y, void* stream) {
  return conv2d_igemm(d, x1, x2, w, bias, res, y, stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                      nullptr);
}

extern "C" int dm_conv2d_gn_fusable(const dm_conv_desc* d, const dm_gn_desc* g) {
  return d && g && g_dm_use_patch_kernel && g_dm_conv_gn_fuse && !d->transposed && dm_conv_patch_gn_ok(d, g);
}

extern "C" int dm_conv2d_igemm_gn(const dm_conv_desc* d, const void* x1, const void* x2, const void* w,
                                  const float* bias, const void* res, void* y, const dm_gn_desc* g, const float* gamma,
                                  const float* beta, const float* scale_shift, const float* pre_add, void* y_norm,
                                  float* mean, float* rstd, void* stream) {
  DM_REQUIRE(g && gamma && beta && y_norm, DM_ERR_ARG, "dm_conv2d_igemm_gn: null argument");
  DM_REQUIRE((mean == nullptr) == (rstd == nullptr), DM_ERR_ARG, "dm_conv2d_igemm_gn: mean and rstd go together");
  DM_REQUIRE((g->has_ss != 0) == (scale_shift != nullptr) && (g->has_pre != 0) == (pre_add != nullptr), DM_ERR_ARG,
             "dm_conv2d_igemm_gn: scale_shift/pre_add pointers disagree with the descriptor");
  DM_REQUIRE(dm_conv2d_gn_fusable(d, g), DM_ERR_UNSUPPORTED,
             "dm_conv2d_igemm_gn: this layer's tiles do not hold whole samples and groups (ask dm_conv2d_gn_fusable first)");
  DM_REQUIRE((uintptr_t)y_norm % 16 == 0, DM_ERR_ARG, "dm_conv2d_igemm_gn: pointers must be 16-byte aligned");
  return conv2d_igemm(d, x1, x2, w, bias, res, y, stream, g, gamma, beta, scale_shift, pre_add, y_norm, mean, rstd);
}

static int conv2d_igemm(const dm_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                        const void* res, void* y, void* stream, const dm_gn_desc* g, const float* gamma,
                        const float* beta, const float* ss, const float* pre, void* y_norm, float* mean,
                        float* rstd, const dm_conv_proj* proj) {
  DM_REQUIRE(d && x1 && w && (y || g), DM_ERR_ARG, "dm_conv2d_igemm: null argument");
  DM_REQUIRE(d->C1 > 0 && d->C1 % BK == 0 && d->C2 >= 0 && d->C2 % BK == 0, DM_ERR_SHAPE,
             "dm_conv2d_igemm: input channels (%d, %d) must be multiples of %d", d->C1, d->C2, BK);
  DM_REQUIRE((d->C2 == 0) == (x2 == nullptr), DM_ERR_ARG, "dm_conv2d_igemm: x2 and C2 disagree");
  DM_REQUIRE(d->Cout > 0 && d->Cout % 8 == 0, DM_ERR_SHAPE, "dm_conv2d_igemm: Cout=%d must be a multiple of 8", d->Cout);
  DM_REQUIRE(d->stride == 1 || d->stride == 2, DM_ERR_UNSUPPORTED, "dm_conv2d_igemm: stride %d", d->stride);
  DM_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->KH <= 7 && d->KW <= 7 && d->pad >= 0, DM_ERR_SHAPE, "dm_conv2d_igemm: kernel %dx%d pad %d", d->KH, d->KW, d->pad);
  DM_REQUIRE(!(d->ups && (d->transposed || d->stride != 1)), DM_ERR_UNSUPPORTED, "dm_conv2d_igemm: fused upsample needs a regular stride-1 conv");
  DM_REQUIRE(d->B > 0 && d->Hs > 0 && d->Ws > 0 && d->Ho > 0 && d->Wo > 0, DM_ERR_SHAPE, "dm_conv2d_igemm: empty tensor");
  const int Hi = d->Hs << d->ups, Wi = d->Ws << d->ups;
  if (d->transposed) {
    // output is the gradient w.r.t. the input of a conv that produced [Hs x Ws]: (Ho-1)/stride... consistency
    DM_REQUIRE((d->Ho + 2 * d->pad - d->KH) / d->stride + 1 == d->Hs && (d->Wo + 2 * d->pad - d->KW) / d->stride + 1 == d->Ws,
               DM_ERR_SHAPE, "dm_conv2d_igemm: transposed dims: out %dx%d does not map to src %dx%d", d->Ho, d->Wo, d->Hs, d->Ws);
  } else {
    DM_REQUIRE((Hi + 2 * d->pad - d->KH) / d->stride + 1 == d->Ho && (Wi + 2 * d->pad - d->KW) / d->stride + 1 == d->Wo,
               DM_ERR_SHAPE, "dm_conv2d_igemm: in %dx%d k%d s%d p%d does not give out %dx%d", Hi, Wi, d->KH, d->stride, d->pad, d->Ho, d->Wo);
  }
  const long npix_src = (long)d->B * d->Hs * d->Ws;
  const long M = (long)d->B * d->Ho * d->Wo;
  const int Cin = d->C1 + d->C2;
  const long wbytes = (long)d->Cout * d->KH * d->KW * Cin * 2;
  DM_REQUIRE(npix_src * d->C1 * 2 < 0x7fffffffL && npix_src * d->C2 * 2 < 0x7fffffffL && wbytes < 0x7fffffffL && M < 0x7fffffffL,
             DM_ERR_UNSUPPORTED, "dm_conv2d_igemm: tensor larger than 2 GiB");

  DM_REQUIRE(((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)res | (uintptr_t)y) % 16 == 0,
             DM_ERR_ARG, "dm_conv2d_igemm: pointers must be 16-byte aligned");
  if (g_dm_timing_twice) {  // timing experiments only: the launch below runs a second time
    const int bit = (d->KH == 1 && d->KW == 1) ? 1 : (d->stride == 2 ? 32 : ((d->KH == 3 && d->stride == 1) ? 128 : 0));
    static thread_local bool again = false;
    if ((g_dm_timing_twice & bit) && !again) {
      again = true;
      conv2d_igemm(d, x1, x2, w, bias, res, y, stream, g, gamma, beta, ss, pre, y_norm, mean, rstd, proj);
      again = false;
    }
  }
  if (g_dm_timing_skip) {   // timing experiments only
    if ((g_dm_timing_skip & 1) && d->KH == 1 && d->KW == 1) return DM_OK;
    if ((g_dm_timing_skip & 32) && d->stride == 2) return DM_OK;
    if ((g_dm_timing_skip & 128) && d->KH == 3 && d->stride == 1) return DM_OK;
  }
  // 3x3 stride-1 convs (and their data gradients) go to the LDS-staged-patch kernel when the shape allows
  if (g_dm_use_patch_kernel &&
      dm_conv_patch_try(d, x1, x2, w, bias, res, y, (hipStream_t)stream, g, gamma, beta, ss, pre, y_norm, mean,
                        rstd, proj) == DM_OK) {
    DM_CHECK_LAUNCH();
    return DM_OK;
  }
  DM_REQUIRE(!g && !proj, DM_ERR_UNSUPPORTED, "dm_conv2d_igemm_gn/_proj: layer not covered by the patch kernel");
  // 1x1 stride-1 layers (and their data gradients) stream their activations past staged weights
  if (dm_conv1x1_try(d, x1, x2, w, bias, res, y, (hipStream_t)stream) == DM_OK) {
    DM_CHECK_LAUNCH();
    return DM_OK;
  }

  ConvArgs a;
  a.x1 = x1; a.x2 = x2; a.w = w; a.bias = bias; a.res = res; a.y = y;
  a.C1 = d->C1; a.C2 = d->C2; a.Cin = Cin; a.Cout = d->Cout;
  a.B = d->B; a.Hs = d->Hs; a.Ws = d->Ws; a.Hi = Hi; a.Wi = Wi; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.ups = d->ups; a.transposed = d->transposed;
  a.M = (int)M;
  a.x1_bytes = (unsigned)(npix_src * d->C1 * 2);
  a.x2_bytes = (unsigned)(npix_src * d->C2 * 2);
  a.w_bytes = (unsigned)wbytes;
  a.n_mtiles = dm_cdiv(M, BM);
  a.subpix = 0; a.tpc = 0; a.Mq = 0;
  if (g_dm_dgrad_subpixel && d->transposed && d->stride == 2 && d->KH == 3 && d->KW == 3 && d->pad == 1 &&
      d->Ho % 2 == 0 && d->Wo % 2 == 0) {
    a.subpix = 1;
    a.Mq = (int)(M / 4);
    a.tpc = dm_cdiv(a.Mq, BM);
    a.n_mtiles = 4 * a.tpc;
  }
  const int mt_pad = (a.n_mtiles + 7) / 8 * 8;
  hipStream_t st = (hipStream_t)stream;
  // 128-wide tiles, unless the output is narrow or 128-wide tiles would leave CUs without a workgroup
  const bool few = (long)a.n_mtiles * dm_cdiv(d->Cout, 128) < 200 && d->Cout % 64 == 0;
  if ((d->Cout % 128 == 0 || d->Cout > 128) && !few) {
    a.n_ntiles = dm_cdiv(d->Cout, 128);
    hipLaunchKernelGGL(conv_igemm_kernel<128>, dim3(mt_pad * a.n_ntiles), dim3(NTHREADS), 0, st, a);
  } else {
    a.n_ntiles = dm_cdiv(d->Cout, 64);
    hipLaunchKernelGGL(conv_igemm_kernel<64>, dim3(mt_pad * a.n_ntiles), dim3(NTHREADS), 0, st, a);
  }
  DM_CHECK_LAUNCH();
  return DM_OK;
}
