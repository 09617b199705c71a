// 1x1 stride-1 convolution (forward, and data gradient of such a conv) as a streaming GEMM on bf16 MFMA:
//
//   y[m][n] = sum_k x[m][k] * w[n][k] + bias[n] + res[m][n]        x = (x1 | x2) along k
//
// These layers (a block's residual projection, the attention block's token projections) move 2-6 bytes per MFMA flop:
// they are bound by the traffic of their activations, not by the matrix cores, and the gathered implicit-GEMM kernel
// (two 32-KB LDS stages, im2col address arithmetic per piece) reaches 3.3 TB/s on them. Here the ACTIVATIONS never go
// through LDS: a lane fetches its MFMA operand - 8 consecutive channels of one pixel, 16 bytes - straight from global
// memory, one chunk of 128 channels ahead of its use; only the weights (re-read by every pixel tile, L2-resident)
// are staged, 32 KB per 128-channel chunk, double buffered. A workgroup is 256 threads and 64 KB of LDS, so two fit on
// a CU, and it leaves most of a CU's registers and LDS to whatever the other stream runs.
//
// The product is formed transposed (weights as the MFMA's row operand, pixels as its columns), and the weight rows
// of a 128-channel tile are dealt to the eight 16-row MFMA tiles such that a lane ends up with 32 CONSECUTIVE output
// channels of one pixel in each of four places: the output leaves the registers as four 16-byte stores per lane, each
// store instruction writing 64 contiguous bytes per pixel, without a pass through LDS.
#include "dm_common.h"

namespace {

constexpr int BM = 128, BN = 128, KC = 128, NT = 256;

struct C1xArgs {
  const void* x1;
  const void* x2;
  const void* w;
  const float* bias;
  const void* res;
  void* y;
  int C1, C2, K, N, M;
  int n_mtiles, n_ntiles;
  unsigned w_bytes;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// row L of the staged weight tile holds output channel perm(L) of the tile: MFMA tile j = L / 16, row m = L % 16 of
// it -> channel (j / 2) * 32 + (m / 4) * 8 + (j % 2) * 4 + m % 4, so that the accumulator registers r of tiles 2g and
// 2g + 1 in lane group q (rows 4q + r) are the 8 channels g * 32 + q * 8 ..: one store instruction writes 64
// contiguous bytes per pixel (its four lane groups side by side), four of them the pixel's 256
__device__ __forceinline__ int perm_row(int L) {
  const int j = L >> 4, m = L & 15;
  return (j >> 1) * 32 + (m >> 2) * 8 + (j & 1) * 4 + (m & 3);
}

__global__ __launch_bounds__(NT, 2) void conv1x1_kernel(C1xArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BN * KC * 2];
  int mt, nt;
  dm_xcd_tile(blockIdx.x, a.n_ntiles, mt, nt);
  if (mt >= a.n_mtiles) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;

  __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
  // weight staging: piece id = tid + 256 i -> LDS row L = id / 16, 16-byte slot id % 16, which holds k-piece
  // slot ^ (L % 16) of the row (fragment reads of 16 rows at one k offset then touch 16 different bank groups)
  unsigned wsrc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int id = tid + NT * i, L = id >> 4, slot = id & 15;
    wsrc[i] = (unsigned)(((n0 + perm_row(L)) * a.K) * 2 + ((slot ^ (L & 15)) << 4));
  }
  auto issue_w = [&](int c, int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      char* dst = smem + buf * (BN * KC * 2) + (wave * 64 + NT * i) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, DM_LDS(dst), 16, wsrc[i] + (unsigned)(c * KC * 2), 0, 0, 0);
    }
  };
  // activation fragments: this wave's pixels m0 + wave * 32 + t * 16 + p (t = 0, 1), channels 8 q .. 8 q + 7 of each
  // 32-channel step; rows beyond M are clamped (computed, never stored)
  int pix[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) pix[t] = min(m0 + wave * 32 + t * 16 + p, a.M - 1);
  auto load_x = [&](int c, bf16x8_t (&xf)[2][4]) {
    const int k0 = c * KC;
    const bool second = k0 >= a.C1;
    const bf16_t* src = (const bf16_t*)(second ? a.x2 : a.x1);
    const int cs = second ? a.C2 : a.C1, kk = (second ? k0 - a.C1 : k0) + q * 8;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xf[t][ks] = *(const bf16x8_t*)(src + (size_t)pix[t] * cs + kk + ks * 32);
  };

  f32x4_t acc[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = a.K / KC;
  bf16x8_t xa[2][4], xb[2][4];
  issue_w(0, 0);
  load_x(0, xa);
  auto compute = [&](int buf, bf16x8_t (&xf)[2][4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16x8_t wf = *(const bf16x8_t*)(smem + buf * (BN * KC * 2) + (j * 16 + p) * (KC * 2) + (((ks * 4 + q) ^ p) << 4));
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[t][ks], acc[t][j], 0, 0, 0);
      }
    }
  };
  for (int c = 0; c < nk; c += 2) {
    // chunk c (buffer 0, fragments xa), then chunk c + 1 (buffer 1, fragments xb): everything this chunk needs has
    // landed, and every wave has left the buffer that the next chunk's weights go into
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (c + 1 < nk) { issue_w(c + 1, 1); load_x(c + 1, xb); }
    compute(0, xa);
    if (c + 1 >= nk) break;
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (c + 2 < nk) { issue_w(c + 2, 0); load_x(c + 2, xa); }
    compute(1, xb);
  }

  // ---- epilogue, straight from the registers: lane (p, q) holds channels n0 + 32 g + 8 q .. + 7 (g = 0..3) of its
  // two pixels
  bf16_t* y = (bf16_t*)a.y;
  const bf16_t* res = (const bf16_t*)a.res;
  const int cbase = n0 + q * 8;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int m = m0 + wave * 32 + t * 16 + p;
    if (m >= a.M) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {   // 8 channels: MFMA tiles j = 2 g, 2 g + 1
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[t][2 * g][r]; v[4 + r] = acc[t][2 * g + 1][r]; }
      const int cn = cbase + g * 32;
      if (a.bias) {
        const f32x4_t b0 = *(const f32x4_t*)(a.bias + cn), b1 = *(const f32x4_t*)(a.bias + cn + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += b0[r]; v[4 + r] += b1[r]; }
      }
      if (res) {
        const bf16x8_t rv = *(const bf16x8_t*)(res + (size_t)m * a.N + cn);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
      }
      *(bf16x8_t*)(y + (size_t)m * a.N + cn) = dm_pack8(v);
    }
  }
}

}  // namespace

extern int g_dm_conv1x1_stream;

// DM_OK if the layer was launched here, DM_ERR_UNSUPPORTED (no error text) if the shape is not covered and the caller
// should use the gathered kernel. Arguments as checked by dm_conv2d_igemm.
int dm_conv1x1_try(const dm_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                   const void* res, void* y, hipStream_t st) {
  if (!g_dm_conv1x1_stream || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->ups) return DM_ERR_UNSUPPORTED;
  if (d->Hs != d->Ho || d->Ws != d->Wo) return DM_ERR_UNSUPPORTED;
  const int K = d->C1 + d->C2;
  if (d->C1 % KC || d->C2 % KC || d->Cout % BN) return DM_ERR_UNSUPPORTED;
  const long M = (long)d->B * d->Ho * d->Wo;
  // Where it pays (g_dm_conv1x1_stream == 1; 2: wherever the shape allows): many pixels and at most two 128-channel
  // chunks of input - the layers whose time is their output's write traffic (a residual projection's data gradient
  // on 32x32 / 16x16 images at batch 128: 34.6 against 44.3 us for 128 -> 384 channels at 32x32). With more chunks a
  // workgroup waits for each chunk's weights behind a barrier (one chunk in flight), and with few pixels there are
  // too few workgroups to cover that: 512 -> 512 at 8x8 takes 18-20 us here against 16-18 us gathered.
  if (g_dm_conv1x1_stream == 1 && (M < 32768 || K > 2 * KC)) return DM_ERR_UNSUPPORTED;
  if (M * K * 2 >= 0x7fffffffL || M * d->Cout * 2 >= 0x7fffffffL) return DM_ERR_UNSUPPORTED;
  C1xArgs a;
  a.x1 = x1; a.x2 = x2; a.w = w; a.bias = bias; a.res = res; a.y = y;
  a.C1 = d->C1; a.C2 = d->C2; a.K = K; a.N = d->Cout; a.M = (int)M;
  a.n_mtiles = dm_cdiv(M, BM);
  a.n_ntiles = d->Cout / BN;
  a.w_bytes = (unsigned)((long)d->Cout * K * 2);
  const int mt_pad = (a.n_mtiles + 7) / 8 * 8;
  hipLaunchKernelGGL(conv1x1_kernel, dim3(mt_pad * a.n_ntiles), dim3(NT), 0, st, a);
  return DM_OK;
}
