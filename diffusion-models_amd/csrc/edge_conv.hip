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
template <int Cs>   // image channels: compile-time, so the accumulators are registers and the channel loops unroll
__global__ __launch_bounds__(NT) void big_to_small_kernel(const bf16_t* __restrict__ big, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int B,
                                                          int H, int W, int Cb, int pix_per_block) {
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
    float acc[Cs];
#pragma unroll
    for (int co = 0; co < Cs; ++co) acc[co] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = h + kh - 1, iw = x + kw - 1;
        if (ih < 0 || iw < 0 || ih >= H || iw >= W) continue;
        float f[8];
        dm_unpack8(*(const bf16x8_t*)(big + ((size_t)(b * H + ih) * W + iw) * Cb + 8 * v), f);
#pragma unroll
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
#pragma unroll
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
template <int CS>
__global__ __launch_bounds__(NT) void edge_wgrad_kernel(const float* __restrict__ small, const bf16_t* __restrict__ big,
                                                        float* __restrict__ part, int want_bias, int B, int H,
                                                        int W, int Cb, int big_is_output, int pix_per_block) {
  constexpr int K = CS * 9;
  constexpr int KP = (K + 1 + 3) / 4 * 4;  // taps + a column of ones (bias), padded to float4
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* S = lds;                 // [WG_PIX][KP]
  float* Bt = lds + WG_PIX * KP;  // [WG_PIX][Cb]
  const int sg = big_is_output ? 1 : -1;
  const int M = B * H * W;
  const int NPG = NT / Cb;        // pixel groups: thread = (feature channel cb, pixel group pg)
  const int cb = threadIdx.x % Cb, pg = threadIdx.x / Cb;
  float acc[KP];
#pragma unroll
  for (int i = 0; i < KP; ++i) acc[i] = 0.f;
  const int m_begin = blockIdx.x * pix_per_block;
  const int m_end = min(M, m_begin + pix_per_block);
  for (int m0 = m_begin; m0 < m_end; m0 += WG_PIX) {
    __syncthreads();
    for (int i = threadIdx.x; i < WG_PIX * KP; i += NT) {
      const int px = i / KP, k = i - px * KP;
      const int m = m0 + px;
      float val = 0.f;
      if (m < m_end) {
        if (k == K) val = 1.f;
        else if (k < K) {
          const int cs = k / 9, tap = k - cs * 9, kh = tap / 3, kw = tap - kh * 3;
          const int b = m / (H * W), r = m - b * H * W, h = r / W, x = r - h * W;
          const int ih = h + sg * (kh - 1), iw = x + sg * (kw - 1);
          if (ih >= 0 && iw >= 0 && ih < H && iw < W) val = small[((size_t)b * CS + cs) * H * W + ih * W + iw];
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
    if (pg < NPG) {
      for (int px = pg; px < WG_PIX; px += NPG) {
        const float bv = Bt[px * Cb + cb];
        const f32x4_t* sp = (const f32x4_t*)(S + px * KP);  // same address in every lane: an LDS broadcast
#pragma unroll
        for (int q = 0; q < KP / 4; ++q) {
          const f32x4_t sv = sp[q];
          acc[4 * q + 0] += bv * sv[0];
          acc[4 * q + 1] += bv * sv[1];
          acc[4 * q + 2] += bv * sv[2];
          acc[4 * q + 3] += bv * sv[3];
        }
      }
    }
  }
  // this workgroup's partial sums: the pixel groups of a channel are added in group order through LDS, then plain
  // stores into the workgroup's row of the partial buffer (laid out like dw, the bias terms behind it)
  float* mine = part + (size_t)blockIdx.x * (K * Cb + (want_bias ? (big_is_output ? Cb : CS) : 0));
  {
    float* red = lds;   // [NT]
#pragma unroll
    for (int k = 0; k <= K; ++k) {
      __syncthreads();
      red[threadIdx.x] = acc[k];
      __syncthreads();
      if (pg == 0) {
        float t = 0.f;
        for (int q = 0; q < NPG; ++q) t += red[q * Cb + cb];
        if (k < K) {
          const int cs = k / 9, tap = k - cs * 9;
          const size_t o = big_is_output ? ((size_t)(cb * CS + cs) * 9 + tap) : ((size_t)(cs * Cb + cb) * 9 + tap);
          mine[o] = t;
        } else if (big_is_output && want_bias) {
          mine[K * Cb + cb] = t;
        }
      }
    }
  }
  if (!big_is_output && want_bias) {
    // out_conv bias: sum of the image-side gradient itself = centre tap (kh = kw = 1) of each channel
    __syncthreads();
    float* red = lds;
    float part[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) part[cs] = 0.f;
    for (int m = m_begin + threadIdx.x; m < m_end; m += NT) {
      const int b = m / (H * W), r = m - b * H * W;
#pragma unroll
      for (int cs = 0; cs < CS; ++cs) part[cs] += small[((size_t)b * CS + cs) * H * W + r];
    }
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
      const float t = dm_wave_sum(part[cs]);
      if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * CS + cs] = t;
    }
    __syncthreads();
    if (threadIdx.x < CS) mine[K * Cb + threadIdx.x] = red[threadIdx.x] + red[CS + threadIdx.x] + red[2 * CS + threadIdx.x] + red[3 * CS + threadIdx.x];
  }
}

// ------------------------------------------------------------------------------------------------
// The same gradients on the matrix cores, for feature widths 64 / 128 and image widths that are multiples of 8:
//   D[cb][k] += sum_m big[m][cb] * S[m][k]     (k = (cs, tap), plus a column of ones for the in_conv bias)
// is a skinny GEMM with the pixels as reduction axis. Per 64-pixel step the feature rows go to LDS as they lie in
// memory ([pixel][cb], 32-byte units XOR-swizzled by row) and are read with the transposing ds_read_b64_tr_b16;
// S is built k-major ([k][pixel], bf16 - the precision the image side has under autocast) so its fragments are
// plain 16-byte reads. One workgroup = 4 waves x (Cb/4 feature channels) over a run of pixels; each workgroup leaves
// its partial sums in its own row of a scratch buffer and an ordered sum adds the rows to dw (no float atomics).
template <int CS, int CB>
__global__ __launch_bounds__(NT) void edge_wgrad_mfma_kernel(const float* __restrict__ small, const bf16_t* __restrict__ big,
                                                             float* __restrict__ part, int want_bias, int B, int H,
                                                             int W, int big_is_output, int pix_per_block) {
  constexpr int K = CS * 9;
  constexpr int NTILE = (K + 1 + 15) / 16;      // 16-column tiles of D
  constexpr int MI = CB / 64;                   // 16-row tiles per wave
  constexpr int PPR = CB / 8, ROWB = CB * 2;    // 16-byte pieces / bytes per feature row
  constexpr int SP = 72;                        // pitch of an S row in elements (64 pixels + 8: bank spread)
  __shared__ __attribute__((aligned(16))) char sbig[64 * ROWB];
  __shared__ __attribute__((aligned(16))) bf16_t sS[NTILE * 16 * SP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sg = big_is_output ? 1 : -1;
  const int HW = H * W, M = B * HW;
  const int m_begin = blockIdx.x * pix_per_block;
  const int m_end = min(M, m_begin + pix_per_block);
  auto swz = [](int row) { return PPR == 16 ? (((row & 3) | ((row >> 1) & 4)) << 1) : ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1); };

  f32x4_t acc[MI][NTILE];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NTILE; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses (as in the conv weight-gradient kernel)
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  int a_off[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int colbase = wave * (CB / 4) + i * 16;
    const int piece = (colbase >> 3) + (p >> 1);
    const int r0 = 8 * g + q;
    a_off[i] = r0 * ROWB + ((piece ^ swz(r0)) << 4) + (p & 1) * 8;
  }

  for (int m0 = m_begin; m0 < m_end; m0 += 64) {
    __syncthreads();
    // feature rows -> LDS
    for (int i = tid; i < 64 * PPR; i += NT) {
      const int row = i / PPR, slot = i - row * PPR;
      const int m = m0 + row;
      s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < m_end) v = *(const s16x8_t*)(big + (size_t)m * CB + 8 * slot);
      *(s16x8_t*)(sbig + row * ROWB + ((slot ^ swz(row)) << 4)) = v;
    }
    // S, k-major: one thread = 8 consecutive pixels (one image row segment) of one k
    for (int i = tid; i < NTILE * 16 * 8; i += NT) {
      const int k = i >> 3, pg = i & 7;
      const int m = m0 + 8 * pg;
      bf16_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
      if (m < m_end) {
        if (k == K) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (bf16_t)1.f;
        } else if (k < K) {
          const int cs = k / 9, tap = k - cs * 9, kh = tap / 3, kw = tap - kh * 3;
          const int b = m / HW, r = m - b * HW, h = r / W, x0 = r - h * W;
          const int ih = h + sg * (kh - 1), iw0 = x0 + sg * (kw - 1);
          if (ih >= 0 && ih < H) {
            const float* src = small + ((size_t)b * CS + cs) * HW + ih * W;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int iw = iw0 + j;
              if (iw >= 0 && iw < W) v[j] = (bf16_t)src[iw];
            }
          }
        }
      }
      *(bf16x8_t*)(sS + k * SP + 8 * pg) = *(const bf16x8_t*)v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[MI], bfr[NTILE];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const char* p0 = sbig + a_off[i] + kk * 32 * ROWB;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 4 * ROWB));
        af[i] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int j = 0; j < NTILE; ++j) bfr[j] = *(const bf16x8_t*)(sS + (j * 16 + li) * SP + kk * 32 + 8 * g);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NTILE; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  // D: lane holds column k = 16 j + (lane & 15), rows cb = wave * CB/4 + 16 i + 4 (lane >> 4) + r; every element has
  // one owner, so the workgroup's partial sums leave with plain stores (laid out like dw, the bias terms behind it)
  float* mine = part + (size_t)blockIdx.x * (K * CB + (want_bias ? (big_is_output ? CB : CS) : 0));
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NTILE; ++j) {
      const int k = j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cb = wave * (CB / 4) + i * 16 + g * 4 + r;
        if (k < K) {
          const int cs = k / 9, tap = k - cs * 9;
          const size_t o = big_is_output ? ((size_t)(cb * CS + cs) * 9 + tap) : ((size_t)(cs * CB + cb) * 9 + tap);
          mine[o] = acc[i][j][r];
        } else if (k == K && big_is_output && want_bias) {
          mine[K * CB + cb] = acc[i][j][r];
        }
      }
    }
  if (!big_is_output && want_bias) {
    // out_conv bias: sum of the image-side gradient itself
    __shared__ float red[4 * CS];
    float part[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) part[cs] = 0.f;
    for (int m = m_begin + tid; m < m_end; m += NT) {
      const int b = m / HW, r = m - b * HW;
#pragma unroll
      for (int cs = 0; cs < CS; ++cs) part[cs] += small[((size_t)b * CS + cs) * HW + r];
    }
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
      const float t = dm_wave_sum(part[cs]);
      if (lane == 0) red[wave * CS + cs] = t;
    }
    __syncthreads();
    if (tid < CS) mine[K * CB + tid] = red[tid] + red[CS + tid] + red[2 * CS + tid] + red[3 * CS + tid];
  }
}

// pixels per workgroup of the two weight-gradient kernels
bool edge_wgrad_on_mfma(int W, int Cb) { return (Cb == 64 || Cb == 128) && W % 8 == 0; }
int edge_wgrad_ppb(long M, int W, int Cb) {
  if (edge_wgrad_on_mfma(W, Cb)) {
    int ppb = 512;
    while (ppb > 64 && dm_cdiv(M, ppb) < 256) ppb /= 2;
    return ppb;
  }
  int ppb = 1024;
  while (ppb > WG_PIX && dm_cdiv(M, ppb) < 128) ppb /= 2;
  return ppb;
}
int edge_wgrad_veclen(int Cs, int Cb, int big_is_output, bool bias) {
  return Cs * 9 * Cb + (bias ? (big_is_output ? Cb : Cs) : 0);
}

template <int CS>
void launch_edge_wgrad_mfma(const float* small, const bf16_t* big, float* part, int want_bias, int B, int H, int W, int Cb,
                            int big_is_output, int ppb, hipStream_t st) {
  const int M = B * H * W;
  dim3 grid(dm_cdiv(M, ppb)), block(NT);
  if (Cb == 128) hipLaunchKernelGGL((edge_wgrad_mfma_kernel<CS, 128>), grid, block, 0, st, small, big, part, want_bias, B, H, W, big_is_output, ppb);
  else hipLaunchKernelGGL((edge_wgrad_mfma_kernel<CS, 64>), grid, block, 0, st, small, big, part, want_bias, B, H, W, big_is_output, ppb);
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
  const dim3 grid(dm_cdiv(M, ppb));
  const int lds = 9 * Cs * Cb * 4;
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* hb = (const bf16_t*)h;
  switch (Cs) {
    case 1: hipLaunchKernelGGL(big_to_small_kernel<1>, grid, dim3(NT), lds, st, hb, w, bias, y, B, H, W, Cb, ppb); break;
    case 2: hipLaunchKernelGGL(big_to_small_kernel<2>, grid, dim3(NT), lds, st, hb, w, bias, y, B, H, W, Cb, ppb); break;
    case 3: hipLaunchKernelGGL(big_to_small_kernel<3>, grid, dim3(NT), lds, st, hb, w, bias, y, B, H, W, Cb, ppb); break;
    default: hipLaunchKernelGGL(big_to_small_kernel<4>, grid, dim3(NT), lds, st, hb, w, bias, y, B, H, W, Cb, ppb); break;
  }
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" long dm_edge_conv_wgrad_ws_floats(int B, int Cs, int H, int W, int Cb, int big_is_output) {
  if (B <= 0 || Cs <= 0 || H <= 0 || W <= 0 || Cb <= 0) return 0;
  const long M = (long)B * H * W;
  return (long)dm_cdiv(M, edge_wgrad_ppb(M, W, Cb)) * edge_wgrad_veclen(Cs, Cb, big_is_output, true);
}

extern "C" int dm_edge_conv_wgrad(const float* small, const void* big, float* dw, float* dbias, int B, int Cs, int H,
                                  int W, int Cb, int big_is_output, float* ws, dm_sum_item* deferred, void* stream) {
  int rc = check_edge("dm_edge_conv_wgrad", B, Cs, H, W, Cb);
  if (rc) return rc;
  DM_REQUIRE(small && big && dw && ws, DM_ERR_ARG, "dm_edge_conv_wgrad: null argument");
  DM_REQUIRE(Cb <= 256, DM_ERR_UNSUPPORTED, "dm_edge_conv_wgrad: feature width %d > 256", Cb);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* bg = (const bf16_t*)big;
  const int M = B * H * W;
  const int ppb = edge_wgrad_ppb(M, W, Cb), nblocks = dm_cdiv(M, ppb);
  const int want_bias = dbias != nullptr;
  const int n = edge_wgrad_veclen(Cs, Cb, big_is_output, want_bias);
  if (edge_wgrad_on_mfma(W, Cb)) {
    switch (Cs) {
      case 1: launch_edge_wgrad_mfma<1>(small, bg, ws, want_bias, B, H, W, Cb, big_is_output, ppb, st); break;
      case 2: launch_edge_wgrad_mfma<2>(small, bg, ws, want_bias, B, H, W, Cb, big_is_output, ppb, st); break;
      case 3: launch_edge_wgrad_mfma<3>(small, bg, ws, want_bias, B, H, W, Cb, big_is_output, ppb, st); break;
      default: launch_edge_wgrad_mfma<4>(small, bg, ws, want_bias, B, H, W, Cb, big_is_output, ppb, st); break;
    }
    DM_CHECK_LAUNCH();
  } else {
    const int kp = (Cs * 9 + 1 + 3) / 4 * 4;
    const int lds = (WG_PIX * kp + WG_PIX * Cb) * 4;
    DM_REQUIRE(lds <= 65536, DM_ERR_UNSUPPORTED, "dm_edge_conv_wgrad: feature width %d exceeds 64 KiB of LDS", Cb);
    dim3 grid(nblocks), block(NT);
    switch (Cs) {
      case 1: hipLaunchKernelGGL(edge_wgrad_kernel<1>, grid, block, lds, st, small, bg, ws, want_bias, B, H, W, Cb, big_is_output, ppb); break;
      case 2: hipLaunchKernelGGL(edge_wgrad_kernel<2>, grid, block, lds, st, small, bg, ws, want_bias, B, H, W, Cb, big_is_output, ppb); break;
      case 3: hipLaunchKernelGGL(edge_wgrad_kernel<3>, grid, block, lds, st, small, bg, ws, want_bias, B, H, W, Cb, big_is_output, ppb); break;
      default: hipLaunchKernelGGL(edge_wgrad_kernel<4>, grid, block, lds, st, small, bg, ws, want_bias, B, H, W, Cb, big_is_output, ppb); break;
    }
    DM_CHECK_LAUNCH();
  }
  const int ndw = Cs * 9 * Cb;
  DmOrderedSum a = {ws, n, nblocks, n, ndw, n, dw, dbias, nullptr, nullptr};
  return dm_finish_ordered_sum(a, deferred, st);
}
