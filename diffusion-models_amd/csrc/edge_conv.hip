// Edge convolutions: the 3x3 'same' convs at the two ends of the UNet, where one side has 1-4 channels
// (image, NCHW fp32) and the other a full feature width (NHWC bf16). With K = 27 (or N = 3) these do not fill an
// MFMA tile; they touch full-resolution tensors once and are HBM-bound, so they run on the VALU with 16-byte
// accesses on the feature side and the (tiny) weights held in LDS.
#include "dm_common.h"

extern int g_dm_edge_mfma;

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
// Matrix-core forms of the two kernels above. The VALU forms spend ~400 instructions per (pixel, 8 channels) on 216
// FMAs, 54 LDS weight reads and 27 bounds-checked loads and run at a fifth of the HBM rate of their one big tensor.
// Here the small side is an im2col row of K = 9 * Cs <= 36 values per pixel, i.e. one (or two) 16x16x32 k-steps.
// fp32 operands keep their precision: each is split into two bf16 (v = hi + lo, |lo| <= 2^-9 |v|) and the product
// is hi*hi + lo*hi + hi*lo, exact bf16 products summed in fp32 (relative error 2^-16 instead of bf16's 2^-9).
__device__ __forceinline__ void split_bf16(float v, bf16_t& hi, bf16_t& lo) {
  hi = (bf16_t)v;
  lo = (bf16_t)(v - (float)hi);
}

// image -> features. D[i][n] = sum_k A[i][k] B[k][n] with i = feature channel (A: the weights, held in registers for
// the whole kernel), n = pixel (B: the pixel's im2col row, gathered by the lane that owns the pixel: lane l holds
// k = 8 (l / 16) .. + 7 of pixel l % 16); the lane then owns 4 consecutive channels of its pixel (an 8-byte store).
// One workgroup = (pixel chunk, 128-channel chunk); NTC = 16-channel tiles per workgroup (1, 2, 4 or 8).
template <int CS, int NTC>
__global__ __launch_bounds__(NT) void small_to_big_mfma_kernel(const float* __restrict__ small, const float* __restrict__ w,
                                                               const float* __restrict__ bias, bf16_t* __restrict__ big,
                                                               int B, int H, int W, int Cb, int sg, int w_big_major,
                                                               int pix_per_block) {
  constexpr int K = CS * 9, KS = (K + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int c0 = blockIdx.y * (NTC * 16);
  const int HW = H * W, M = B * HW;
  bf16x8_t ahi[KS][NTC], alo[KS][NTC];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      const int cb = c0 + 16 * t + li;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = ks * 32 + 8 * g + e;
        float v = 0.f;
        if (k < K && cb < Cb) {
          const int cs = k / 9, tap = k - cs * 9;
          v = w_big_major ? w[(cb * CS + cs) * 9 + tap] : w[(cs * Cb + cb) * 9 + tap];
        }
        bf16_t h, l;
        split_bf16(v, h, l);
        ahi[ks][t][e] = h;
        alo[ks][t][e] = l;
      }
    }
  f32x4_t bv[NTC];
#pragma unroll
  for (int t = 0; t < NTC; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cb = c0 + 16 * t + 4 * g + r;
      bv[t][r] = (bias && cb < Cb) ? bias[cb] : 0.f;
    }
  // this lane's 8 (or 16) im2col entries: channel plane offset, row / column step (0 beyond K)
  int koff[KS][8], kdh[KS][8], kdw[KS][8];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ks * 32 + 8 * g + e;
      const int cs = k / 9, tap = k - cs * 9, kh = tap / 3, kw = tap - kh * 3;
      kdh[ks][e] = k < K ? sg * (kh - 1) : 4;   // 4: never inside the image's 3x3 neighbourhood test below
      kdw[ks][e] = sg * (kw - 1);
      koff[ks][e] = cs * HW;
    }
  const int m_begin = blockIdx.x * pix_per_block, m_end = min(M, m_begin + pix_per_block);
  for (int m0 = m_begin + 16 * wave; m0 < m_end; m0 += 64) {
    const int m = m0 + li;
    const bool live = m < m_end;
    const int mm = live ? m : m_end - 1;
    const int b = mm / HW, r0 = mm - b * HW, h = r0 / W, x = r0 - h * W;
    const float* sp = small + (size_t)b * CS * HW;
    bf16x8_t bhi[KS], blo[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ih = h + kdh[ks][e], iw = x + kdw[ks][e];
        float v = 0.f;
        if (kdh[ks][e] != 4 && ih >= 0 && ih < H && iw >= 0 && iw < W) v = sp[koff[ks][e] + ih * W + iw];
        bf16_t hh, ll;
        split_bf16(v, hh, ll);
        bhi[ks][e] = hh;
        blo[ks][e] = ll;
      }
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      f32x4_t acc = bv[t];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[ks][t], bhi[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[ks][t], blo[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[ks][t], bhi[ks], acc, 0, 0, 0);
      }
      const int cb = c0 + 16 * t + 4 * g;
      if (live && cb < Cb) {
        const bf16x4_t o = {(bf16_t)acc[0], (bf16_t)acc[1], (bf16_t)acc[2], (bf16_t)acc[3]};
        *(bf16x4_t*)(big + (size_t)m * Cb + cb) = o;
      }
    }
  }
}

// features -> image. D[i][n] with i = image channel (A: the weights as hi/lo bf16 fragments in LDS, rows >= Cs are
// zero), n = pixel (B: 16 bytes of the feature row of the pixel's tap neighbour, straight from global memory - each
// row is read by 9 taps and comes from cache for 8 of them). k-steps: (tap, 32-channel block). Lanes 0..15 end up
// with the Cs outputs of their pixel: consecutive pixels of one image plane, a coalesced fp32 store per channel.
template <int CS>
__global__ __launch_bounds__(NT) void big_to_small_mfma_kernel(const bf16_t* __restrict__ big, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int B,
                                                               int H, int W, int Cb, int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) char wl_raw[];   // [kstep][hi | lo][co * 4 + g] fragments of 16 bytes
  bf16x8_t* wl = (bf16x8_t*)wl_raw;
  const int nblk = Cb >> 5, nks = 9 * nblk;
  for (int i = threadIdx.x; i < nks * CS * 4; i += NT) {
    const int s = i / (CS * 4), rr = i - s * (CS * 4), co = rr >> 2, gg = rr & 3;
    const int tap = s / nblk, blk = s - tap * nblk;
    bf16x8_t hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bf16_t hh, ll;
      split_bf16(w[(co * Cb + blk * 32 + 8 * gg + e) * 9 + tap], hh, ll);
      hi[e] = hh;
      lo[e] = ll;
    }
    wl[(s * 2 + 0) * (CS * 4) + rr] = hi;
    wl[(s * 2 + 1) * (CS * 4) + rr] = lo;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int HW = H * W, M = B * HW;
  const bool arow = li < CS;
  const int aslot = (arow ? li : 0) * 4 + g;
  const bf16x8_t zero8 = {};
  const int m_begin = blockIdx.x * pix_per_block, m_end = min(M, m_begin + pix_per_block);
  for (int m0 = m_begin + 32 * wave; m0 < m_end; m0 += 128) {   // two groups of 16 pixels per wave and round
    int pb[2], ph[2], px[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + 16 * u + li;
      live[u] = m < m_end;
      const int mm = live[u] ? m : m_end - 1;
      pb[u] = mm / HW;
      const int r0 = mm - pb[u] * HW;
      ph[u] = r0 / W;
      px[u] = r0 - ph[u] * W;
    }
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    // k-steps in batches of 4 32-channel blocks: the batch's 8 feature loads are issued together (a wave has nothing
    // else to hide their latency under), then its 16 MFMAs run
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      const bf16_t* src[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ih = ph[u] + kh - 1, iw = px[u] + kw - 1;
        const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
        src[u] = ok ? big + ((size_t)(pb[u] * H + ih) * W + iw) * Cb + 8 * g : nullptr;
      }
      for (int blk0 = 0; blk0 < nblk; blk0 += 4) {
        bf16x8_t f[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            f[q][u] = (src[u] && blk0 + q < nblk) ? *(const bf16x8_t*)(src[u] + (blk0 + q) * 32) : zero8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int s = tap * nblk + min(blk0 + q, nblk - 1);   // (a block beyond the end multiplies zeros)
          const bf16x8_t a_hi = arow ? wl[(s * 2 + 0) * (CS * 4) + aslot] : zero8;
          const bf16x8_t a_lo = arow ? wl[(s * 2 + 1) * (CS * 4) + aslot] : zero8;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, f[q][u], acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, f[q][u], acc[u], 0, 0, 0);
          }
        }
      }
    }
    if (g == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (live[u]) {
#pragma unroll
          for (int co = 0; co < CS; ++co)
            y[((size_t)pb[u] * CS + co) * HW + ph[u] * W + px[u]] = acc[u][co] + (bias ? bias[co] : 0.f);
        }
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

template <int CS>
void launch_small_to_big_mfma(const float* small, const float* w, const float* bias, bf16_t* big, int B, int H, int W, int Cb,
                              int sg, int w_big_major, hipStream_t st) {
  const int M = B * H * W, ppb = 512;
  const int ntc = Cb >= 128 ? 8 : Cb / 16;
  const dim3 grid(dm_cdiv(M, ppb), dm_cdiv(Cb, ntc * 16));
  switch (ntc) {
    case 8: hipLaunchKernelGGL((small_to_big_mfma_kernel<CS, 8>), grid, dim3(NT), 0, st, small, w, bias, big, B, H, W, Cb, sg, w_big_major, ppb); break;
    case 4: hipLaunchKernelGGL((small_to_big_mfma_kernel<CS, 4>), grid, dim3(NT), 0, st, small, w, bias, big, B, H, W, Cb, sg, w_big_major, ppb); break;
    case 2: hipLaunchKernelGGL((small_to_big_mfma_kernel<CS, 2>), grid, dim3(NT), 0, st, small, w, bias, big, B, H, W, Cb, sg, w_big_major, ppb); break;
    default: hipLaunchKernelGGL((small_to_big_mfma_kernel<CS, 1>), grid, dim3(NT), 0, st, small, w, bias, big, B, H, W, Cb, sg, w_big_major, ppb); break;
  }
}

void launch_small_to_big(const float* small, const float* w, const float* bias, bf16_t* big, int B, int Cs, int H, int W,
                         int Cb, int sg, int w_big_major, hipStream_t st) {
  if (g_dm_edge_mfma && Cb % 16 == 0) {
    switch (Cs) {
      case 1: launch_small_to_big_mfma<1>(small, w, bias, big, B, H, W, Cb, sg, w_big_major, st); break;
      case 2: launch_small_to_big_mfma<2>(small, w, bias, big, B, H, W, Cb, sg, w_big_major, st); break;
      case 3: launch_small_to_big_mfma<3>(small, w, bias, big, B, H, W, Cb, sg, w_big_major, st); break;
      default: launch_small_to_big_mfma<4>(small, w, bias, big, B, H, W, Cb, sg, w_big_major, st); break;
    }
    return;
  }
  const int M = B * H * W, ppb = 128;
  hipLaunchKernelGGL(small_to_big_kernel, dim3(dm_cdiv(M, ppb)), dim3(NT), Cs * 9 * Cb * 4, st, small, w, bias, big, B, Cs,
                     H, W, Cb, sg, w_big_major, ppb);
}

}  // namespace

extern "C" int dm_in_conv_fwd(const float* x, const float* w, const float* bias, void* y, int B, int Cs, int H, int W,
                              int Cb, void* stream) {
  int rc = check_edge("dm_in_conv_fwd", B, Cs, H, W, Cb);
  if (rc) return rc;
  DM_REQUIRE(x && w && y, DM_ERR_ARG, "dm_in_conv_fwd: null argument");
  launch_small_to_big(x, w, bias, (bf16_t*)y, B, Cs, H, W, Cb, 1, 1, (hipStream_t)stream);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_out_conv_dgrad(const float* dy, const float* w, void* dh, int B, int Cs, int H, int W, int Cb,
                                 void* stream) {
  int rc = check_edge("dm_out_conv_dgrad", B, Cs, H, W, Cb);
  if (rc) return rc;
  DM_REQUIRE(dy && w && dh, DM_ERR_ARG, "dm_out_conv_dgrad: null argument");
  launch_small_to_big(dy, w, nullptr, (bf16_t*)dh, B, Cs, H, W, Cb, -1, 0, (hipStream_t)stream);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_out_conv_fwd(const void* h, const float* w, const float* bias, float* y, int B, int Cs, int H, int W,
                               int Cb, void* stream) {
  int rc = check_edge("dm_out_conv_fwd", B, Cs, H, W, Cb);
  if (rc) return rc;
  DM_REQUIRE(h && w && y, DM_ERR_ARG, "dm_out_conv_fwd: null argument");
  const int M = B * H * W;
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* hb = (const bf16_t*)h;
  if (g_dm_edge_mfma && Cb % 32 == 0) {
    // 128 pixels per workgroup (32 per wave): several workgroups per CU hide each other's load latency
    const int ppb = 128, lds = 9 * (Cb / 32) * 2 * Cs * 4 * 16;   // at most 72 KiB (Cs = 4, Cb = 512)
    const dim3 grid(dm_cdiv(M, ppb));
    switch (Cs) {
      case 1: hipLaunchKernelGGL(big_to_small_mfma_kernel<1>, grid, dim3(NT), lds, st, hb, w, bias, y, B, H, W, Cb, ppb); break;
      case 2: hipLaunchKernelGGL(big_to_small_mfma_kernel<2>, grid, dim3(NT), lds, st, hb, w, bias, y, B, H, W, Cb, ppb); break;
      case 3: hipLaunchKernelGGL(big_to_small_mfma_kernel<3>, grid, dim3(NT), lds, st, hb, w, bias, y, B, H, W, Cb, ppb); break;
      default: hipLaunchKernelGGL(big_to_small_mfma_kernel<4>, grid, dim3(NT), lds, st, hb, w, bias, y, B, H, W, Cb, ppb); break;
    }
    DM_CHECK_LAUNCH();
    return DM_OK;
  }
  const int ppb = 256;
  DM_REQUIRE(9 * Cs * Cb * 4 <= 65536, DM_ERR_UNSUPPORTED, "dm_out_conv_fwd: %d x %d weights exceed 64 KiB of LDS", Cs, Cb);
  const dim3 grid(dm_cdiv(M, ppb));
  const int lds = 9 * Cs * Cb * 4;
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
