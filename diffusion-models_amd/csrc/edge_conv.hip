// Edge convolutions: the 3x3 'same' convs at the two ends of the UNet, where one side has 1-4 channels
// (image, NCHW fp32) and the other a full feature width (NHWC bf16). With K = 27 (or N = 3) these do not fill an
// MFMA tile; they touch full-resolution tensors once and are HBM-bound, so they run on the VALU with 16-byte
// accesses on the feature side and the (tiny) weights held in LDS.
#include "dm_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXCS = 4;    // channels on the image side
constexpr int MAXCB = 512;  // channels on the feature side

// ------------------------------------------------------------------------------------------------
// image -> features:  big[m][cb] = bias[cb] + sum_{cs,tap} small[b][cs][h + sg*(kh-1)][w + sg*(kw-1)] * W(cb,cs,tap)
//   w_big_major = 1: W = w[cb][cs][tap] (in_conv forward, sg = +1)
//   w_big_major = 0: W = w[cs][cb][tap] (out_conv data gradient, sg = -1)
__global__ __launch_bounds__(NT) void small_to_big_kernel(const float* __restrict__ small, const float* __restrict__ w,
                                                          const float* __restrict__ bias, bf16_t* __restrict__ big,
                                                          int B, int Cs, int H, int W, int Cb, int sg, int w_big_major,
                                                          int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cs*9][Cb]
  const int K = Cs * 9;
  for (int i = threadIdx.x; i < K * Cb; i += NT) {
    int k = i / Cb, cb = i - k * Cb;
    int cs = k / 9, tap = k - cs * 9;
    wl[i] = w_big_major ? w[(cb * Cs + cs) * 9 + tap] : w[(cs * Cb + cb) * 9 + tap];
  }
  __syncthreads();
  const int VP = Cb >> 3, PL = NT / VP;
  const int v = threadIdx.x % VP, pl = threadIdx.x / VP;
  if (pl >= PL) return;
  const int M = B * H * W;
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = bias ? bias[8 * v + j] : 0.f;
  const int m_end = min(M, (int)(blockIdx.x + 1) * pix_per_block);
  for (int m = blockIdx.x * pix_per_block + pl; m < m_end; m += PL) {
    const int b = m / (H * W), r = m - b * H * W, h = r / W, x = r - h * W;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bv[j];
    for (int cs = 0; cs < Cs; ++cs) {
      const float* sp = small + ((size_t)b * Cs + cs) * H * W;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ih = h + sg * (kh - 1), iw = x + sg * (kw - 1);
          if (ih < 0 || iw < 0 || ih >= H || iw >= W) continue;
          const float s = sp[ih * W + iw];
          const float* wp = wl + (cs * 9 + kh * 3 + kw) * Cb + 8 * v;
          const f32x4_t w0 = *(const f32x4_t*)wp, w1 = *(const f32x4_t*)(wp + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[j] += s * w0[j]; acc[4 + j] += s * w1[j]; }
        }
      }
    }
    *(bf16x8_t*)(big + (size_t)m * Cb + 8 * v) = dm_pack8(acc);
  }
}

// ------------------------------------------------------------------------------------------------
// features -> image (out_conv forward): y[b][co][h][w] = bias[co] + sum_{tap,cb} big[(h+kh-1,w+kw-1)][cb] * w[co][cb][tap]
// Cb/8 lanes share one pixel (16 bytes of the feature row each) and reduce with shuffles.
__global__ __launch_bounds__(NT) void big_to_small_kernel(const bf16_t* __restrict__ big, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int B,
                                                          int Cs, int H, int W, int Cb, int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [9][Cs][Cb]
  for (int i = threadIdx.x; i < 9 * Cs * Cb; i += NT) {
    int tap = i / (Cs * Cb), r = i - tap * Cs * Cb, co = r / Cb, cb = r - co * Cb;
    wl[i] = w[(co * Cb + cb) * 9 + tap];
  }
  __syncthreads();
  const int VP = Cb >> 3, PL = NT / VP;  // VP is a power of two <= 64
  const int v = threadIdx.x % VP, pl = threadIdx.x / VP;
  const int M = B * H * W;
  const int m_end = min(M, (int)(blockIdx.x + 1) * pix_per_block);
  // every lane of a wave runs the same trip count so the shuffles below see all their partners
  for (int m0 = blockIdx.x * pix_per_block; m0 < m_end; m0 += PL) {
    const int m = m0 + pl;
    const bool ok = m < m_end;
    const int mm = ok ? m : m_end - 1;
    const int b = mm / (H * W), r = mm - b * H * W, h = r / W, x = r - h * W;
    float acc[MAXCS] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = h + kh - 1, iw = x + kw - 1;
        if (ih < 0 || iw < 0 || ih >= H || iw >= W) continue;
        float f[8];
        dm_unpack8(*(const bf16x8_t*)(big + ((size_t)(b * H + ih) * W + iw) * Cb + 8 * v), f);
        for (int co = 0; co < Cs; ++co) {
          const float* wp = wl + ((kh * 3 + kw) * Cs + co) * Cb + 8 * v;
          const f32x4_t w0 = *(const f32x4_t*)wp, w1 = *(const f32x4_t*)(wp + 4);
          float t = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) t += f[j] * w0[j] + f[4 + j] * w1[j];
          acc[co] += t;
        }
      }
    }
    for (int co = 0; co < Cs; ++co) {
      float t = acc[co];
      for (int o = VP >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if (ok && v == 0) y[((size_t)b * Cs + co) * H * W + h * W + x] = t + (bias ? bias[co] : 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight / bias gradients: out(cb, k) += sum_m big[m][cb] * S_k(m),  k = (cs, tap),
//   S_k(m) = small[b][cs][h + sg*(kh-1)][w + sg*(kw-1)]  (0 outside the image)
// big_is_output = 1 (in_conv):  dw[cb][cs][tap], sg = +1, dbias[cb] += sum_m big[m][cb]
// big_is_output = 0 (out_conv): dw[cs][cb][tap], sg = -1, dbias[cs] += sum_m small[b][cs][h][w]
constexpr int WG_PIX = 64;
__global__ __launch_bounds__(NT) void edge_wgrad_kernel(const float* __restrict__ small, const bf16_t* __restrict__ big,
                                                        float* __restrict__ dw, float* __restrict__ dbias, int B, int Cs,
                                                        int H, int W, int Cb, int big_is_output, int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int K = Cs * 9;
  float* S = lds;                     // [K+1][WG_PIX]  (last row: ones / unused)
  float* Bt = lds + (K + 1) * WG_PIX;  // [WG_PIX][Cb]
  const int sg = big_is_output ? 1 : -1;
  const int M = B * H * W;
  const int KG = NT / Cb;             // k-groups (Cb <= 256 so KG >= 1)
  const int cb = threadIdx.x % Cb, kg = threadIdx.x / Cb;
  const bool active = kg < KG;
  float acc[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = 0.f;
  float bsum = 0.f;  // per-thread bias partial
  const int m_begin = blockIdx.x * pix_per_block;
  const int m_end = min(M, m_begin + pix_per_block);
  for (int m0 = m_begin; m0 < m_end; m0 += WG_PIX) {
    __syncthreads();
    for (int i = threadIdx.x; i < (K + 1) * WG_PIX; i += NT) {
      const int k = i / WG_PIX, px = i - k * WG_PIX;
      const int m = m0 + px;
      float val = 0.f;
      if (m < m_end) {
        if (k == K) val = 1.f;
        else {
          const int cs = k / 9, tap = k - cs * 9, kh = tap / 3, kw = tap - kh * 3;
          const int b = m / (H * W), r = m - b * H * W, h = r / W, x = r - h * W;
          const int ih = h + sg * (kh - 1), iw = x + sg * (kw - 1);
          if (ih >= 0 && iw >= 0 && ih < H && iw < W) val = small[((size_t)b * Cs + cs) * H * W + ih * W + iw];
        }
      }
      S[i] = val;
    }
    for (int i = threadIdx.x; i < WG_PIX * (Cb >> 3); i += NT) {
      const int px = i / (Cb >> 3), v = i - px * (Cb >> 3);
      const int m = m0 + px;
      float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < m_end) dm_unpack8(*(const bf16x8_t*)(big + (size_t)m * Cb + 8 * v), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) Bt[px * Cb + 8 * v + j] = f[j];
    }
    __syncthreads();
    if (active) {
      for (int px = 0; px < WG_PIX; ++px) {
        const float bvv = Bt[px * Cb + cb];
#pragma unroll
        for (int i = 0; i < 28; ++i) {
          const int k = kg + i * KG;
          if (k <= K) acc[i] += bvv * S[k * WG_PIX + px];
        }
      }
    }
    if (!big_is_output && threadIdx.x < Cs) {
      // centre tap (kh = kw = 1) of channel cs is the unshifted image value
      for (int px = 0; px < WG_PIX; ++px) bsum += S[(threadIdx.x * 9 + 4) * WG_PIX + px];
    }
  }
  if (active) {
#pragma unroll
    for (int i = 0; i < 28; ++i) {
      const int k = kg + i * KG;
      if (k < K) {
        const int cs = k / 9, tap = k - cs * 9;
        const size_t o = big_is_output ? ((size_t)(cb * Cs + cs) * 9 + tap) : ((size_t)(cs * Cb + cb) * 9 + tap);
        atomicAdd(dw + o, acc[i]);
      } else if (k == K && big_is_output && dbias) {
        atomicAdd(dbias + cb, acc[i]);
      }
    }
  }
  if (!big_is_output && dbias && threadIdx.x < Cs) atomicAdd(dbias + threadIdx.x, bsum);
}

int check_edge(const char* who, int B, int Cs, int H, int W, int Cb) {
  DM_REQUIRE(B > 0 && H > 0 && W > 0, DM_ERR_SHAPE, "%s: empty tensor", who);
  DM_REQUIRE(Cs >= 1 && Cs <= MAXCS, DM_ERR_UNSUPPORTED, "%s: image side has %d channels (max %d)", who, Cs, MAXCS);
  DM_REQUIRE(Cb >= 8 && Cb <= MAXCB && (Cb & (Cb - 1)) == 0, DM_ERR_UNSUPPORTED,
             "%s: feature width %d must be a power of two in [8, %d]", who, Cb, MAXCB);
  DM_REQUIRE((long)B * H * W * Cb * 2 < 0x7fffffffL, DM_ERR_UNSUPPORTED, "%s: tensor larger than 2 GiB", who);
  return DM_OK;
}

}  // namespace

extern "C" int dm_in_conv_fwd(const float* x, const float* w, const float* bias, void* y, int B, int Cs, int H, int W,
                              int Cb, void* stream) {
  int rc = check_edge("dm_in_conv_fwd", B, Cs, H, W, Cb);
  if (rc) return rc;
  DM_REQUIRE(x && w && y, DM_ERR_ARG, "dm_in_conv_fwd: null argument");
  const int M = B * H * W, ppb = 128;
  hipLaunchKernelGGL(small_to_big_kernel, dim3(dm_cdiv(M, ppb)), dim3(NT), Cs * 9 * Cb * 4, (hipStream_t)stream, x, w, bias,
                     (bf16_t*)y, B, Cs, H, W, Cb, 1, 1, ppb);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_out_conv_dgrad(const float* dy, const float* w, void* dh, int B, int Cs, int H, int W, int Cb,
                                 void* stream) {
  int rc = check_edge("dm_out_conv_dgrad", B, Cs, H, W, Cb);
  if (rc) return rc;
  DM_REQUIRE(dy && w && dh, DM_ERR_ARG, "dm_out_conv_dgrad: null argument");
  const int M = B * H * W, ppb = 128;
  hipLaunchKernelGGL(small_to_big_kernel, dim3(dm_cdiv(M, ppb)), dim3(NT), Cs * 9 * Cb * 4, (hipStream_t)stream, dy, w,
                     (const float*)nullptr, (bf16_t*)dh, B, Cs, H, W, Cb, -1, 0, ppb);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_out_conv_fwd(const void* h, const float* w, const float* bias, float* y, int B, int Cs, int H, int W,
                               int Cb, void* stream) {
  int rc = check_edge("dm_out_conv_fwd", B, Cs, H, W, Cb);
  if (rc) return rc;
  DM_REQUIRE(h && w && y, DM_ERR_ARG, "dm_out_conv_fwd: null argument");
  const int M = B * H * W, ppb = 256;
  DM_REQUIRE(9 * Cs * Cb * 4 <= 65536, DM_ERR_UNSUPPORTED, "dm_out_conv_fwd: %d x %d weights exceed 64 KiB of LDS", Cs, Cb);
  hipLaunchKernelGGL(big_to_small_kernel, dim3(dm_cdiv(M, ppb)), dim3(NT), 9 * Cs * Cb * 4, (hipStream_t)stream,
                     (const bf16_t*)h, w, bias, y, B, Cs, H, W, Cb, ppb);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_edge_conv_wgrad(const float* small, const void* big, float* dw, float* dbias, int B, int Cs, int H,
                                  int W, int Cb, int big_is_output, void* stream) {
  int rc = check_edge("dm_edge_conv_wgrad", B, Cs, H, W, Cb);
  if (rc) return rc;
  DM_REQUIRE(small && big && dw, DM_ERR_ARG, "dm_edge_conv_wgrad: null argument");
  DM_REQUIRE(Cb <= 256, DM_ERR_UNSUPPORTED, "dm_edge_conv_wgrad: feature width %d > 256", Cb);
  DM_REQUIRE((Cs * 9 + 1 + (NT / Cb) - 1) / (NT / Cb) <= 28, DM_ERR_UNSUPPORTED, "dm_edge_conv_wgrad: too many taps per thread");
  const int M = B * H * W, ppb = 256;
  const int lds = ((Cs * 9 + 1) * WG_PIX + WG_PIX * Cb) * 4;
  DM_REQUIRE(lds <= 65536, DM_ERR_UNSUPPORTED, "dm_edge_conv_wgrad: feature width %d exceeds 64 KiB of LDS", Cb);
  hipLaunchKernelGGL(edge_wgrad_kernel, dim3(dm_cdiv(M, ppb)), dim3(NT), lds, (hipStream_t)stream, small,
                     (const bf16_t*)big, dw, dbias, B, Cs, H, W, Cb, big_is_output, ppb);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
