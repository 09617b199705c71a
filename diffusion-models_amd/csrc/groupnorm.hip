// Fused GroupNorm (+ per-sample scale/shift or pre-add) (+ SiLU), forward and backward, NHWC bf16.
//
// HBM-bound streaming design. Every kernel walks whole pixel rows (all C channels of a pixel are contiguous, so
// accesses are 16-byte pieces of full rows) with (sample, pixel-chunk) workgroups, i.e. thousands of independent
// workgroups regardless of the tensor shape:
//   forward   1. per-(sample, channel) sum and sum of squares -> a [B][2][C] workspace; one workgroup per
//                (sample, 64 channels) walks all pixels, so the sums are plain stores in a fixed order: the
//                forward pass is bit-reproducible
//             2. (sample, pixel-chunk) workgroups derive their sample's group statistics from that workspace (a few
//                hundred flops) and stream y = act(ka[c] * x + kb[c])
//   backward  1. per-(sample, channel) sums of du, du * xhat (and xhat), same decomposition as forward 1
//             2. group sums from the workspace, then dx streamed; the first pixel-chunk of each sample also emits
//                the per-sample parameter-side terms (dscale_shift, dpre_add, and the sample's contribution to
//                dgamma/dbeta/upstream bias into a [B][3][C] buffer that dm_groupnorm_param_grads sums over the
//                batch in a fixed order - no float atomics, so the gradients are bitwise reproducible)
// The second sweep of each pair re-reads a tensor that was just read, which the 256 MiB Infinity Cache serves.
// The workspace is pure scratch (written before it is read in every call), so one buffer serves every call.
// The channel axis may be the concatenation of two tensors (x1 | x2).
#include "dm_common.h"
#include <mutex>

extern int g_dm_gn_fused;
extern int g_dm_gn_bwd16;
extern int g_dm_gn_min_ppt;

// Scratch of the entry point that allocates nothing (dm_groupnorm_act_fwd_noalloc): a fixed device array that
// stands in for the caller-provided workspace. Every GPU the library runs on holds its own copy at its own
// address; host code must not use the array's name as a pointer - it resolves the address once per device with
// hipGetSymbolAddress (scratch_for_current_device) and passes that to the kernels.
constexpr long NOALLOC_WS_FLOATS = 1L << 20;   // 4 MiB per GPU: covers B * C <= 349525 (e.g. batch 256 x 1024 channels)
static __device__ __align__(256) float scratch[NOALLOC_WS_FLOATS];

namespace {

constexpr int NT = 256;
constexpr int MAXC = 1024;

struct GnArgs {
  const bf16_t* x1;
  const bf16_t* x2;
  const float* gamma;
  const float* beta;
  const float* ss;
  const float* pre;
  const bf16_t* dy;
  const bf16_t* dextra;
  const bf16_t* dextra2;   // a second gradient to add (that of a skip connection's late consumer), or null
  bf16_t* y;
  bf16_t* dx1;
  bf16_t* dx2;
  float* mean;
  float* rstd;
  float* dss;
  float* dpre;
  float* pp;    // per-sample parameter-gradient terms [B][3][C]: dbeta, dgamma, sum of dx over pixels (when want_dup)
  int want_dup; // the third term (bias gradient of the convolution that produced x) is asked for
  float* ws;
  int B, HW, C1, C2, C, G, cg, act, pch;
  int sw;   // channels per workgroup of the single-pass kernels: whole groups, a multiple of 8, at most 64 (0: none)
  float eps;
};

__device__ __forceinline__ bf16x8_t load_x(const GnArgs& a, size_t pix, int c) {
  if (c < a.C1) return *(const bf16x8_t*)(a.x1 + pix * a.C1 + c);
  return *(const bf16x8_t*)(a.x2 + pix * a.C2 + (c - a.C1));
}

// thread -> (8-channel vector v, pixel lane pl); threads beyond PL*VP idle
struct Lanes {
  int VP, PL, v, pl;
  bool active;
  __device__ Lanes(int C) {
    VP = C >> 3;
    PL = NT / VP;
    v = threadIdx.x % VP;
    pl = threadIdx.x / VP;
    active = pl < PL;
  }
};

// ---- sweep 1 of both directions: one workgroup = (sample b, 64 channels starting at c0 = 64 * blockIdx.x).
// Thread = (8-channel vector v = tid & 7, pixel lane pl = tid >> 3); every pixel contributes one 128-byte segment.
constexpr int SL_C = 64, SL_V = SL_C / 8, SL_PL = NT / SL_V;   // 64 channels, 8 vectors, 32 pixel lanes

// sums of NV per-thread 8-vectors over the 32 pixel lanes, in a fixed order, stored to out[k * C + c]
template <int NV>
__device__ __forceinline__ void slab_reduce_store(float (&part)[NV][8], float* red, int C, int c0, float* out) {
  const int v = threadIdx.x & (SL_V - 1), pl = threadIdx.x / SL_V;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(pl * NV + k) * SL_C + v * 8 + j] = part[k][j];
  __syncthreads();
  for (int i = threadIdx.x; i < NV * SL_C; i += NT) {
    const int k = i / SL_C, cc = i - k * SL_C;
    float s = 0.f;
    for (int q = 0; q < SL_PL; ++q) s += red[(q * NV + k) * SL_C + cc];
    if (c0 + cc < C) out[(size_t)k * C + c0 + cc] = s;
  }
}

// ---- forward sweep 1: ws[b][0][c] = sum x, ws[b][1][c] = sum x^2
__global__ __launch_bounds__(NT) void gn_stats_kernel(GnArgs a) {
  __shared__ float red[SL_PL * 2 * SL_C];
  const int b = blockIdx.y, c0 = blockIdx.x * SL_C;
  const int v = threadIdx.x & (SL_V - 1), pl = threadIdx.x / SL_V;
  const int c = c0 + 8 * v;
  float part[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[0][j] = part[1][j] = 0.f;
  if (c < a.C) {
    const size_t base = (size_t)b * a.HW;
    int p = pl;
    for (; p + 3 * SL_PL < a.HW; p += 4 * SL_PL) {
      bf16x8_t r0 = load_x(a, base + p, c), r1 = load_x(a, base + p + SL_PL, c);
      bf16x8_t r2 = load_x(a, base + p + 2 * SL_PL, c), r3 = load_x(a, base + p + 3 * SL_PL, c);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f0 = (float)r0[j], f1 = (float)r1[j], f2 = (float)r2[j], f3 = (float)r3[j];
        part[0][j] += (f0 + f1) + (f2 + f3);
        part[1][j] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
      }
    }
    for (; p < a.HW; p += SL_PL) {
      bf16x8_t r0 = load_x(a, base + p, c);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f0 = (float)r0[j]; part[0][j] += f0; part[1][j] += f0 * f0; }
    }
  }
  slab_reduce_store<2>(part, red, a.C, c0, a.ws + (size_t)b * 2 * a.C);
}

// group statistics of sample b from the channel sums (with the pre-add folded in analytically)
__device__ __forceinline__ void group_stats(const GnArgs& a, int b, float* s_mean, float* s_rstd, bool write_out) {
  const float* w1 = a.ws + (size_t)b * 2 * a.C;
  const float* w2 = w1 + a.C;
  for (int g = threadIdx.x; g < a.G; g += NT) {
    float t1 = 0.f, t2 = 0.f;
    for (int j = 0; j < a.cg; ++j) {
      const int c = g * a.cg + j;
      float u1 = w1[c], u2 = w2[c];
      if (a.pre) {
        const float pv = a.pre[(size_t)b * a.C + c];
        u2 += 2.f * pv * u1 + (float)a.HW * pv * pv;
        u1 += (float)a.HW * pv;
      }
      t1 += u1;
      t2 += u2;
    }
    const float inv_n = 1.0f / (float)(a.cg * a.HW);
    const float m = t1 * inv_n;
    const float var = fmaxf(t2 * inv_n - m * m, 0.f);
    const float r = rsqrtf(var + a.eps);
    s_mean[g] = m;
    s_rstd[g] = r;
    if (write_out) { a.mean[(size_t)b * a.G + g] = m; a.rstd[(size_t)b * a.G + g] = r; }
  }
}

// ---- forward sweep 2: y = act(ka * x + kb)
__global__ __launch_bounds__(NT) void gn_apply_kernel(GnArgs a) {
  __shared__ float s_mean[256], s_rstd[256];
  __shared__ float s_ka[MAXC], s_kb[MAXC];
  const int b = blockIdx.y, p0 = blockIdx.x * a.pch;
  const int p1 = min(a.HW, p0 + a.pch);
  group_stats(a, b, s_mean, s_rstd, blockIdx.x == 0);
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += NT) {
    const int g = c / a.cg;
    const float r = s_rstd[g], mu = s_mean[g];
    const float ga = a.gamma[c], be = a.beta[c];
    float sc = 1.f, sh = 0.f;
    if (a.ss) { sc = 1.f + a.ss[(size_t)b * 2 * a.C + c]; sh = a.ss[(size_t)b * 2 * a.C + a.C + c]; }
    const float pv = a.pre ? a.pre[(size_t)b * a.C + c] : 0.f;
    const float ka = sc * ga * r;
    s_ka[c] = ka;
    s_kb[c] = ka * (pv - mu) + sc * be + sh;  // u = sc * (ga * (x + pre - mu) * r + be) + sh
  }
  __syncthreads();
  const Lanes L(a.C);
  if (!L.active) return;
  const int c = 8 * L.v;
  float ka[8], kb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ka[j] = s_ka[c + j]; kb[j] = s_kb[c + j]; }
  const size_t base = (size_t)b * a.HW;
  auto emit = [&](const bf16x8_t& r, int p) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float u = ka[j] * (float)r[j] + kb[j];
      f[j] = a.act ? dm_silu(u) : u;
    }
    *(bf16x8_t*)(a.y + (base + p) * a.C + c) = dm_pack8(f);
  };
  int p = p0 + L.pl;
  for (; p + 3 * L.PL < p1; p += 4 * L.PL) {
    bf16x8_t r0 = load_x(a, base + p, c), r1 = load_x(a, base + p + L.PL, c);
    bf16x8_t r2 = load_x(a, base + p + 2 * L.PL, c), r3 = load_x(a, base + p + 3 * L.PL, c);
    emit(r0, p); emit(r1, p + L.PL); emit(r2, p + 2 * L.PL); emit(r3, p + 3 * L.PL);
  }
  for (; p < p1; p += L.PL) emit(load_x(a, base + p, c), p);
}

// per-channel constants of sample b for the backward: xhat = xa * x + xb, u = ua * x + ub, k = sc * gamma, r
__device__ __forceinline__ void bwd_constants(const GnArgs& a, int b, float* s_xa, float* s_xb, float* s_ua, float* s_ub) {
  for (int c = threadIdx.x; c < a.C; c += NT) {
    const int g = c / a.cg;
    const float r = a.rstd[(size_t)b * a.G + g], mu = a.mean[(size_t)b * a.G + g];
    const float pv = a.pre ? a.pre[(size_t)b * a.C + c] : 0.f;
    const float sc = a.ss ? 1.f + a.ss[(size_t)b * 2 * a.C + c] : 1.f;
    const float sh = a.ss ? a.ss[(size_t)b * 2 * a.C + a.C + c] : 0.f;
    const float xa = r, xb = (pv - mu) * r;
    s_xa[c] = xa;
    s_xb[c] = xb;
    s_ua[c] = sc * a.gamma[c] * xa;
    s_ub[c] = sc * (a.gamma[c] * xb + a.beta[c]) + sh;
  }
}

// ---- backward sweep 1: ws[b][0][c] = sum du, ws[b][1][c] = sum du*xhat, ws[b][2][c] = sum xhat
__global__ __launch_bounds__(NT) void gn_bwd_reduce_kernel(GnArgs a) {
  __shared__ float red[SL_PL * 3 * SL_C];
  const int b = blockIdx.y, c0 = blockIdx.x * SL_C;
  const int v = threadIdx.x & (SL_V - 1), pl = threadIdx.x / SL_V;
  const int c = c0 + 8 * v;
  float part[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[0][j] = part[1][j] = part[2][j] = 0.f;
  if (c < a.C) {
    // per-channel constants: xhat = xa * x + xb, u = ua * x + ub
    float xa[8], xb[8], ua[8], ub[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cj = c + j, g = cj / a.cg;
      const float r = a.rstd[(size_t)b * a.G + g], mu = a.mean[(size_t)b * a.G + g];
      const float pv = a.pre ? a.pre[(size_t)b * a.C + cj] : 0.f;
      const float sc = a.ss ? 1.f + a.ss[(size_t)b * 2 * a.C + cj] : 1.f;
      const float sh = a.ss ? a.ss[(size_t)b * 2 * a.C + a.C + cj] : 0.f;
      xa[j] = r;
      xb[j] = (pv - mu) * r;
      ua[j] = sc * a.gamma[cj] * xa[j];
      ub[j] = sc * (a.gamma[cj] * xb[j] + a.beta[cj]) + sh;
    }
    const size_t base = (size_t)b * a.HW;
    auto acc = [&](const bf16x8_t& xr, const bf16x8_t& dr) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = (float)xr[j];
        const float xh = xa[j] * xv + xb[j];
        const float du = a.act ? (float)dr[j] * dm_silu_grad(ua[j] * xv + ub[j]) : (float)dr[j];
        part[0][j] += du;
        part[1][j] += du * xh;
        part[2][j] += xh;
      }
    };
    int p = pl;
    for (; p + SL_PL < a.HW; p += 2 * SL_PL) {
      bf16x8_t x0 = load_x(a, base + p, c), x1 = load_x(a, base + p + SL_PL, c);
      bf16x8_t d0 = *(const bf16x8_t*)(a.dy + (base + p) * a.C + c);
      bf16x8_t d1 = *(const bf16x8_t*)(a.dy + (base + p + SL_PL) * a.C + c);
      acc(x0, d0);
      acc(x1, d1);
    }
    for (; p < a.HW; p += SL_PL) acc(load_x(a, base + p, c), *(const bf16x8_t*)(a.dy + (base + p) * a.C + c));
  }
  slab_reduce_store<3>(part, red, a.C, c0, a.ws + (size_t)b * 3 * a.C);
}

// ---- backward sweep 2: dx = r * (k * du - A_g - xhat * B_g) (+ dextra); parameter-side gradients from chunk 0
__global__ __launch_bounds__(NT) void gn_bwd_apply_kernel(GnArgs a) {
  __shared__ float s_A[256], s_B[256];
  __shared__ float s_xa[MAXC], s_xb[MAXC], s_ua[MAXC], s_ub[MAXC];
  const int b = blockIdx.y, p0 = blockIdx.x * a.pch;
  const int p1 = min(a.HW, p0 + a.pch);
  const float* wa = a.ws + (size_t)b * 3 * a.C;
  const float* wb = wa + a.C;
  const float* wx = wb + a.C;
  bwd_constants(a, b, s_xa, s_xb, s_ua, s_ub);
  const float inv_n = 1.0f / (float)(a.cg * a.HW);
  for (int g = threadIdx.x; g < a.G; g += NT) {
    float A = 0.f, Bv = 0.f;
    for (int j = 0; j < a.cg; ++j) {
      const int c = g * a.cg + j;
      const float k = a.gamma[c] * (a.ss ? 1.f + a.ss[(size_t)b * 2 * a.C + c] : 1.f);
      A += k * wa[c];
      Bv += k * wb[c];
    }
    s_A[g] = A * inv_n;
    s_B[g] = Bv * inv_n;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < a.C; c += NT) {
      const float av = wa[c], bv = wb[c];
      const float sc = a.ss ? 1.f + a.ss[(size_t)b * 2 * a.C + c] : 1.f;
      float* pp = a.pp + (size_t)b * 3 * a.C;
      pp[c] = sc * av;
      pp[a.C + c] = sc * bv;
      if (a.dss) {
        a.dss[(size_t)b * 2 * a.C + c] = a.gamma[c] * bv + a.beta[c] * av;
        a.dss[(size_t)b * 2 * a.C + a.C + c] = av;
      }
      if (a.dpre || a.want_dup) {
        const int g = c / a.cg;
        // sum over pixels of dx = r * (k * a_c - HW * A_g - B_g * sum xhat_c)
        const float dxsum = s_xa[c] * (sc * a.gamma[c] * av - (float)a.HW * s_A[g] - s_B[g] * wx[c]);
        if (a.dpre) a.dpre[(size_t)b * a.C + c] = dxsum;
        if (a.want_dup) pp[2 * a.C + c] = dxsum;
      }
    }
  }
  __syncthreads();
  const Lanes L(a.C);
  if (!L.active) return;
  const int c = 8 * L.v;
  float xa[8], xb[8], ua[8], ub[8], Ag[8], Bg[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xa[j] = s_xa[c + j]; xb[j] = s_xb[c + j]; ua[j] = s_ua[c + j]; ub[j] = s_ub[c + j];
    Ag[j] = s_A[(c + j) / a.cg]; Bg[j] = s_B[(c + j) / a.cg];
  }
  const size_t base = (size_t)b * a.HW;
  auto emit = [&](const bf16x8_t& xr, const bf16x8_t& dr, int p) {
    const size_t o = (base + p) * a.C + c;
    float f[8], e[8], e2[8];
    if (a.dextra) dm_unpack8(*(const bf16x8_t*)(a.dextra + o), e);
    if (a.dextra2) dm_unpack8(*(const bf16x8_t*)(a.dextra2 + o), e2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xv = (float)xr[j];
      const float xh = xa[j] * xv + xb[j];
      const float du = a.act ? (float)dr[j] * dm_silu_grad(ua[j] * xv + ub[j]) : (float)dr[j];
      // k * r = ua (u = ua * x + ub with ua = sc * gamma * r)
      const float dxv = ua[j] * du - xa[j] * (Ag[j] + xh * Bg[j]);
      f[j] = a.dextra ? dxv + e[j] : dxv;
      if (a.dextra2) f[j] += e2[j];
    }
    const bf16x8_t ov = dm_pack8(f);
    if (c < a.C1) *(bf16x8_t*)(a.dx1 + (base + p) * a.C1 + c) = ov;
    else *(bf16x8_t*)(a.dx2 + (base + p) * a.C2 + (c - a.C1)) = ov;
  };
  int p = p0 + L.pl;
  for (; p + L.PL < p1; p += 2 * L.PL) {
    bf16x8_t x0 = load_x(a, base + p, c), x1 = load_x(a, base + p + L.PL, c);
    bf16x8_t d0 = *(const bf16x8_t*)(a.dy + (base + p) * a.C + c);
    bf16x8_t d1 = *(const bf16x8_t*)(a.dy + (base + p + L.PL) * a.C + c);
    emit(x0, d0, p);
    emit(x1, d1, p + L.PL);
  }
  for (; p < p1; p += L.PL) emit(load_x(a, base + p, c), *(const bf16x8_t*)(a.dy + (base + p) * a.C + c), p);
}

// ============================================================================================================
// Single-pass kernels for the common case (at most 1024 pixels per sample; whole groups fit a slab of at most 64
// channels that is a multiple of 8: 64 when the group size divides 64, 48 for groups of 6, 12 or 24 channels - the
// concatenated inputs of the up path - with the last two of a pixel's eight vector lanes idle).
// One workgroup = (sample b, one slab of channels): it holds that slab of x (and dy) IN REGISTERS as packed bf16 - thread =
// (8-channel vector, pixel lane), PPT pixels per thread - so every tensor is read from memory exactly once and the
// sums never leave the CU: forward 4 B/element instead of 6, backward 6 (+2 with a residual gradient) instead of 10.
// PPT is the smallest of 1, 2, 4, 8 that brings the pixel lanes down to 64 (512 threads; 128 lanes = 1024 threads for
// 513..1024 pixels), so small images still fill a workgroup with waves (an 8x8 image is 64 lanes of one pixel, not
// one wave holding eight) and the loads of a whole slab are in flight at once.
// Groups never straddle the slab, so the group statistics are workgroup-local; all sums run in a fixed order.

// Barrier for data handed over through LDS only. __syncthreads() also drains every outstanding global load and STORE
// of the wave (vmcnt(0)); in the single-pass kernels, whose workgroups live for one latency chain, each such drain
// behind a store of statistics or parameter partials is a memory round trip added to the chain.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// sums of NV per-thread 8-vectors over all pixel lanes: shuffles inside a wave (lanes with equal tid & 7), then a
// fixed-order sum over the waves through LDS. Result for channel cc (0..63), quantity k in out[k * 64 + cc].
template <int NV>
__device__ __forceinline__ void slab_block_sum(float (&part)[NV][8], float* red /*[waves][NV][64]*/, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = part[k][j];
      t += __shfl_xor(t, 8, 64);
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      part[k][j] = t;
    }
  if (lane < 8) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(wave * NV + k) * 64 + lane * 8 + j] = part[k][j];
  }
  lds_barrier();
  for (int i = threadIdx.x; i < NV * 64; i += blockDim.x) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += red[w * NV * 64 + i];
    out[i] = t;
  }
  lds_barrier();
}

// element j (0..7) of 8 packed bf16 held as 4 words: a shift or a mask, no conversion instruction; holding the slab
// as words keeps it at 4 registers per vector (a bf16 vector held across the reduction is widened to one register
// per element by the compiler)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ float bf16_half(unsigned u, int hi) {
  return __uint_as_float(hi ? (u & 0xffff0000u) : (u << 16));
}
__device__ __forceinline__ float bf16_at(const u32x4_t& w, int j) {
  const unsigned u = w[j >> 1];
  return __uint_as_float((j & 1) ? (u & 0xffff0000u) : (u << 16));
}

// the slab's vector of pixel p (clamped to a valid address; out-of-range lanes hold zeros)
__device__ __forceinline__ u32x4_t load_slab_x(const GnArgs& a, size_t base, int p, int c, bool cok) {
  const bool ok = cok && p < a.HW;
  const u32x4_t v = __builtin_bit_cast(u32x4_t, load_x(a, base + (ok ? p : 0), cok ? c : 0));
  return ok ? v : u32x4_t{0, 0, 0, 0};
}

// the same sum in two steps, for kernels that produce the 8 channels of a thread as two halves of 4 (the first
// half's sums then need no registers while the second half is being worked on): slab_half_to_lds per half h, then
// slab_sum_finish
template <int NV>
__device__ __forceinline__ void slab_half_to_lds(float (&ph)[NV][4], float* red /*[waves][NV][64]*/, int h) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = ph[k][j];
      t += __shfl_xor(t, 8, 64);
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      ph[k][j] = t;
    }
  if (lane < 8) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave * NV + k) * 64 + lane * 8 + 4 * h + j] = ph[k][j];
  }
}
template <int NV>
__device__ __forceinline__ void slab_sum_finish(const float* red, float* out) {
  const int nw = blockDim.x >> 6;
  lds_barrier();
  for (int i = threadIdx.x; i < NV * 64; i += blockDim.x) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += red[w * NV * 64 + i];
    out[i] = t;
  }
  lds_barrier();
}

template <int PPT, int MAXT, bool ACT>
__global__ __launch_bounds__(MAXT) void gn_fwd_fused_kernel(GnArgs a) {
  __shared__ float red[(MAXT / 64) * 2 * 64];
  __shared__ float sums[2 * 64];
  __shared__ float s_mean[64], s_rstd[64];   // per group of the slab
  __shared__ float s_ka[64], s_kb[64], s_pre[64];
  const int b = blockIdx.y, c0 = blockIdx.x * a.sw;
  const int v = threadIdx.x & 7, pl = threadIdx.x >> 3, PLN = blockDim.x >> 3;
  const int c = c0 + 8 * v;
  const bool cok = 8 * v < a.sw && c < a.C;   // (slabs narrower than 64 channels leave their last vector lanes idle)
  const size_t base = (size_t)b * a.HW;
  u32x4_t xr[PPT];   // packed bf16 pairs: 4 registers per 8 channels of a pixel
  float part[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) part[0][j] = part[1][j] = 0.f;
#pragma unroll
  for (int i = 0; i < PPT; ++i) xr[i] = load_slab_x(a, base, pl + i * PLN, c, cok);
  // this thread's channel constants (threads 0..63: one channel of the slab each), fetched beside the slab instead of
  // behind the reduction
  float c_gamma = 0.f, c_beta = 0.f, c_sc = 1.f, c_sh = 0.f, c_pre = 0.f;
  if (threadIdx.x < 64) {
    const int cj = c0 + (int)threadIdx.x;
    if ((int)threadIdx.x < a.sw && cj < a.C) {
      c_gamma = a.gamma[cj];
      c_beta = a.beta[cj];
      if (a.ss) { c_sc = 1.f + a.ss[(size_t)b * 2 * a.C + cj]; c_sh = a.ss[(size_t)b * 2 * a.C + a.C + cj]; }
      if (a.pre) c_pre = a.pre[(size_t)b * a.C + cj];
    }
    s_pre[threadIdx.x] = c_pre;   // (read behind the barriers of the block sum)
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float f = bf16_at(xr[i], j); part[0][j] += f; part[1][j] += f * f; }
  slab_block_sum<2>(part, red, sums);
#pragma unroll
  for (int i = 0; i < PPT; ++i) asm volatile("" : "+v"(xr[i]));   // keep the slab packed across the reduction
  // group statistics (pre-add folded in analytically)
  const int ng = a.sw / a.cg;
  for (int gl = threadIdx.x; gl < ng; gl += blockDim.x) {
    const int g = c0 / a.cg + gl;
    if (g < a.G) {
      float t1 = 0.f, t2 = 0.f;
      for (int j = 0; j < a.cg; ++j) {
        const int cc = gl * a.cg + j;
        float u1 = sums[cc], u2 = sums[64 + cc];
        if (a.pre) {
          const float pv = s_pre[cc];
          u2 += 2.f * pv * u1 + (float)a.HW * pv * pv;
          u1 += (float)a.HW * pv;
        }
        t1 += u1;
        t2 += u2;
      }
      const float inv_n = 1.0f / (float)(a.cg * a.HW);
      const float m = t1 * inv_n;
      const float var = fmaxf(t2 * inv_n - m * m, 0.f);
      const float r = rsqrtf(var + a.eps);
      s_mean[gl] = m;
      s_rstd[gl] = r;
      a.mean[(size_t)b * a.G + g] = m;
      a.rstd[(size_t)b * a.G + g] = r;
    }
  }
  lds_barrier();
  // per-channel affine of the slab, once per workgroup
  if (threadIdx.x < 64) {
    const int cc = threadIdx.x, cj = c0 + cc;
    float ka = 0.f, kb = 0.f;
    if (cc < a.sw && cj < a.C) {
      const int gl = cc / a.cg;
      const float r = s_rstd[gl], mu = s_mean[gl];
      ka = c_sc * c_gamma * r;
      kb = ka * (c_pre - mu) + c_sc * c_beta + c_sh;
    }
    s_ka[cc] = ka;
    s_kb[cc] = kb;
  }
  lds_barrier();
  if (!cok) return;
  float ka[8], kb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ka[j] = s_ka[8 * v + j]; kb[j] = s_kb[8 * v + j]; }
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = pl + i * PLN;
    if (p < a.HW) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = ka[j] * bf16_at(xr[i], j) + kb[j];
        f[j] = ACT ? dm_silu(u) : u;
      }
      *(bf16x8_t*)(a.y + (base + p) * a.C + c) = dm_pack8(f);
    }
  }
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  return __builtin_bit_cast(unsigned, bf16x2_t{(bf16_t)lo, (bf16_t)hi});
}

// Backward. The kernel is as much VALU- as HBM-bound (the SiLU gradient costs an exponential and a reciprocal per
// element, and a one-workgroup-per-CU kernel cannot hide arithmetic under its own loads), so:
//  - all arithmetic runs on PAIRS of channels (the two bf16 of a 32-bit word) as 2-wide float vectors, which the
//    compiler maps to the packed v_pk_* instructions: half the instruction count;
//  - du = dy * silu'(u) is computed ONCE, in the first pass (fp32, and summed in fp32), then rounded to bf16 into the
//    registers that held dy; the second pass reads it back instead of evaluating the SiLU gradient again. dx thus
//    carries one extra bf16 rounding of one of its two terms (it is itself stored as bf16).
// Register budget: the slab (x and dy, 8 * PPT registers) plus per-channel state; the 8 channels of a thread are
// processed as two halves of 4 so that only 4 channels' constants are live at a time.
__device__ __forceinline__ f32x2_t bf16_pair(unsigned w) {
  return f32x2_t{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}
__device__ __forceinline__ unsigned pack_pair(f32x2_t v) { return pack_bf16x2(v[0], v[1]); }
__device__ __forceinline__ f32x2_t silu_grad_pair(f32x2_t u) {
  const f32x2_t e = {__expf(-u[0]), __expf(-u[1])};
  const f32x2_t d = e + 1.0f;
  const f32x2_t sg = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return sg * (u * (1.0f - sg) + 1.0f);
}

// EXACT: the slab tiles the workgroup exactly (HW = PPT * pixel lanes, all 8 vector lanes inside C): no masking
template <int PPT, int MAXT, bool ACT, bool EXTRA, bool EXACT>
__global__ __launch_bounds__(MAXT) void gn_bwd_fused_kernel(GnArgs a) {
  __shared__ float red[(MAXT / 64) * 3 * 64];
  __shared__ float sums[3 * 64];
  __shared__ float s_A[64], s_B[64];       // per group of the slab
  __shared__ __attribute__((aligned(16))) float s_k[4][64];   // per channel: xhat = xa * x + xb, u = ua * x + ub
  __shared__ __attribute__((aligned(16))) float s_AB[2][64];  // per channel: A and B of its group
  __shared__ float s_kg[64];               // per channel: gamma * (1 + scale)
  __shared__ float s_c[4][64];             // per channel: 1 + scale, gamma, beta, rstd (for the parameter-side results)
  const int b = blockIdx.y, c0 = blockIdx.x * a.sw;
  const int v = threadIdx.x & 7, pl = threadIdx.x >> 3, PLN = blockDim.x >> 3;
  const int c = c0 + 8 * v;
  const bool cok = 8 * v < a.sw && c < a.C;
  const size_t base = (size_t)b * a.HW;
  // the slab as individual 32-bit registers (a 4-register vector that is updated element by element has to be
  // re-assembled in fresh registers wherever it is used whole, which doubles the footprint at that point)
  unsigned xr[PPT][4], dr[PPT][4];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = pl + i * PLN;
    u32x4_t xv, dv;
    if (EXACT) {
      xv = __builtin_bit_cast(u32x4_t, load_x(a, base + p, c));
      dv = *(const u32x4_t*)(a.dy + (base + p) * a.C + c);
    } else {
      const bool ok = cok && p < a.HW;
      xv = load_slab_x(a, base, p, c, cok);
      dv = *(const u32x4_t*)(a.dy + (base + (ok ? p : 0)) * a.C + (cok ? c : 0));
      if (!ok) dv = u32x4_t{0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { xr[i][k] = xv[k]; dr[i][k] = dv[k]; }
  }
  constexpr bool EARLY = EXTRA && PPT <= 2;
  const bool two = EXTRA && a.dextra2 != nullptr;
  u32x4_t ew[PPT], ew2[PPT];
  auto fetch = [&](int i, int p) {
    const size_t o = (base + (EXACT ? p : min(p, a.HW - 1))) * a.C + (cok ? c : 0);
    ew[i] = *(const u32x4_t*)(a.dextra + o);
    if (two) ew2[i] = *(const u32x4_t*)(a.dextra2 + o);
  };
  if (EARLY) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) fetch(i, pl + i * PLN);
  }
  // threads 0..63 own one channel of the slab each: its constants are fetched here, beside the slab, and kept (in
  // LDS: the 16-pixel variant has no register to spare) for the parameter-side results behind the reduction, which
  // then need no second trip to memory
  if (threadIdx.x < 64) {
    float c_sc = 1.f, c_gamma = 0.f, c_beta = 0.f, c_r = 0.f;
    const int cc = threadIdx.x, cj = c0 + cc;
    float kxa = 0.f, kxb = 0.f, kua = 0.f, kub = 0.f;
    if (cc < a.sw && cj < a.C) {
      const int g = cj / a.cg;
      const float r = a.rstd[(size_t)b * a.G + g], mu = a.mean[(size_t)b * a.G + g];
      const float pv = a.pre ? a.pre[(size_t)b * a.C + cj] : 0.f;
      const float sc = a.ss ? 1.f + a.ss[(size_t)b * 2 * a.C + cj] : 1.f;
      const float sh = a.ss ? a.ss[(size_t)b * 2 * a.C + a.C + cj] : 0.f;
      c_sc = sc; c_gamma = a.gamma[cj]; c_beta = a.beta[cj]; c_r = r;
      kxa = r;
      kxb = (pv - mu) * r;
      kua = sc * c_gamma * r;
      kub = sc * (c_gamma * kxb + c_beta) + sh;
    }
    s_k[0][cc] = kxa; s_k[1][cc] = kxb; s_k[2][cc] = kua; s_k[3][cc] = kub;
    s_kg[cc] = c_gamma * c_sc;
    s_c[0][cc] = c_sc; s_c[1][cc] = c_gamma; s_c[2][cc] = c_beta; s_c[3][cc] = c_r;
  }
  lds_barrier();   // constants visible (the slab's loads may still be in flight)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4_t xa = *(const f32x4_t*)&s_k[0][8 * v + 4 * h], xb = *(const f32x4_t*)&s_k[1][8 * v + 4 * h];
    const f32x4_t ua = *(const f32x4_t*)&s_k[2][8 * v + 4 * h], ub = *(const f32x4_t*)&s_k[3][8 * v + 4 * h];
    f32x2_t p0[2] = {{0.f, 0.f}, {0.f, 0.f}}, p1[2] = {{0.f, 0.f}, {0.f, 0.f}}, p2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float live = (EXACT || pl + i * PLN < a.HW) ? 1.f : 0.f;   // xhat of a lane beyond the image must not be summed
      // an opaque copy of the pixel's words: the unpacking then cannot be hoisted above this point, so the compiler
      // works through the slab one pixel at a time instead of widening all of it to floats first
      unsigned xw[2] = {xr[i][2 * h], xr[i][2 * h + 1]}, dw[2] = {dr[i][2 * h], dr[i][2 * h + 1]};
      asm volatile("" : "+v"(xw[0]), "+v"(xw[1]), "+v"(dw[0]), "+v"(dw[1]));
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x2_t kxa = {xa[2 * q], xa[2 * q + 1]}, kxb = {xb[2 * q], xb[2 * q + 1]};
        const f32x2_t xv = bf16_pair(xw[q]), dv = bf16_pair(dw[q]);
        const f32x2_t xh = kxa * xv + kxb;
        f32x2_t du = dv;
        if (ACT) {
          const f32x2_t kua = {ua[2 * q], ua[2 * q + 1]}, kub = {ub[2 * q], ub[2 * q + 1]};
          du = dv * silu_grad_pair(kua * xv + kub);
        }
        p0[q] += du;
        p1[q] += du * xh;
        p2[q] += EXACT ? xh : xh * live;
        if (ACT) dr[i][2 * h + q] = pack_pair(du);   // dy is consumed: its register takes du for the second pass
      }
    }
    float ph[3][4] = {{p0[0][0], p0[0][1], p0[1][0], p0[1][1]}, {p1[0][0], p1[0][1], p1[1][0], p1[1][1]},
                      {p2[0][0], p2[0][1], p2[1][0], p2[1][1]}};
    slab_half_to_lds<3>(ph, red, h);
  }
  slab_sum_finish<3>(red, sums);
  // make the slab opaque here: otherwise the compiler keeps per-element values of the first pass alive for the second
#pragma unroll
  for (int i = 0; i < PPT; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(xr[i][k]), "+v"(dr[i][k]));
  const float inv_n = 1.0f / (float)(a.cg * a.HW);
  const int ng = a.sw / a.cg;
  for (int gl = threadIdx.x; gl < ng; gl += blockDim.x) {
    float A = 0.f, Bv = 0.f;
    for (int j = 0; j < a.cg; ++j) {
      const int cc = gl * a.cg + j, cj = c0 + cc;
      if (cj < a.C) {
        const float k = s_kg[cc];
        A += k * sums[cc];
        Bv += k * sums[64 + cc];
      }
    }
    s_A[gl] = A * inv_n;
    s_B[gl] = Bv * inv_n;
  }
  lds_barrier();
  // parameter-side gradients of this slab's channels; A and B per channel for the second pass
  if (threadIdx.x < 64) {
    const int cc = threadIdx.x, cj = c0 + cc, gl = cc < a.sw ? cc / a.cg : 0;
    s_AB[0][cc] = s_A[gl];
    s_AB[1][cc] = s_B[gl];
    if (cc < a.sw && cj < a.C) {
      const float av = sums[cc], bv = sums[64 + cc], xs = sums[128 + cc];
      const float c_sc = s_c[0][cc], c_gamma = s_c[1][cc], c_beta = s_c[2][cc], c_r = s_c[3][cc];
      float* pp = a.pp + (size_t)b * 3 * a.C;
      pp[cj] = c_sc * av;
      pp[a.C + cj] = c_sc * bv;
      if (a.dss) {
        a.dss[(size_t)b * 2 * a.C + cj] = c_gamma * bv + c_beta * av;
        a.dss[(size_t)b * 2 * a.C + a.C + cj] = av;
      }
      if (a.dpre || a.want_dup) {
        const float dxsum = c_r * (c_sc * c_gamma * av - (float)a.HW * s_A[gl] - s_B[gl] * xs);
        if (a.dpre) a.dpre[(size_t)b * a.C + cj] = dxsum;
        if (a.want_dup) pp[2 * a.C + cj] = dxsum;
      }
    }
  }
  lds_barrier();   // (the stores above are still on their way: nothing in this kernel reads them)
  if (!cok) return;
  // second pass: dx = ua * du - xa * (A + xhat * B) (+ dextra), pixel by pixel with all 8 channels' constants live
  // (no accumulators are left to hold), stored straight from the registers
  f32x4_t xa[2], xb[2], ua[2], Ag[2], Bg[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    xa[h] = *(const f32x4_t*)&s_k[0][8 * v + 4 * h]; xb[h] = *(const f32x4_t*)&s_k[1][8 * v + 4 * h];
    ua[h] = *(const f32x4_t*)&s_k[2][8 * v + 4 * h];
    Ag[h] = *(const f32x4_t*)&s_AB[0][8 * v + 4 * h]; Bg[h] = *(const f32x4_t*)&s_AB[1][8 * v + 4 * h];
  }
  bf16_t* const dxb = c < a.C1 ? a.dx1 + base * a.C1 + c : a.dx2 + base * a.C2 + (c - a.C1);
  const int dxs = c < a.C1 ? a.C1 : a.C2;
  // The addends are read AHEAD pixels in front of their use (a window of registers, the slab has taken the rest): a
  // load per pixel issued where it is needed would put its full latency into every iteration.
  // (Slabs of up to 2 pixels per thread have the registers to hold all their addends: those were fetched at the
  // top of the kernel, beside the slab, which takes the second trip to memory out of a small layer's latency chain.)
  constexpr int AHEAD = PPT > 2 ? 2 : 1;
  if (EXTRA && !EARLY) {
#pragma unroll
    for (int i = 0; i < AHEAD && i < PPT; ++i) fetch(i, pl + i * PLN);
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    int p = pl + i * PLN;
    unsigned xw[4] = {xr[i][0], xr[i][1], xr[i][2], xr[i][3]}, dw[4] = {dr[i][0], dr[i][1], dr[i][2], dr[i][3]};
    // opaque: unpacking and address arithmetic stay with their pixel instead of being hoisted for the whole slab
    asm volatile("" : "+v"(xw[0]), "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3]), "+v"(dw[0]), "+v"(dw[1]), "+v"(dw[2]), "+v"(dw[3]), "+v"(p));
    if (EXTRA && !EARLY && i + AHEAD < PPT) fetch(i + AHEAD, p + AHEAD * PLN);   // (address from the opaque p: stays here)
    u32x4_t ow;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int h = w >> 1, q = w & 1;
      const f32x2_t kxa = {xa[h][2 * q], xa[h][2 * q + 1]}, kxb = {xb[h][2 * q], xb[h][2 * q + 1]};
      const f32x2_t kua = {ua[h][2 * q], ua[h][2 * q + 1]};
      const f32x2_t kA = {Ag[h][2 * q], Ag[h][2 * q + 1]}, kB = {Bg[h][2 * q], Bg[h][2 * q + 1]};
      const f32x2_t xh = kxa * bf16_pair(xw[w]) + kxb;
      f32x2_t r = kua * bf16_pair(dw[w]) - kxa * (kA + xh * kB);
      if (EXTRA) r += bf16_pair(ew[i][w]);
      if (two) r += bf16_pair(ew2[i][w]);
      ow[w] = pack_pair(r);
    }
    if (EXACT || p < a.HW) *(u32x4_t*)(dxb + (size_t)p * dxs) = ow;
  }
}

// the single-pass kernels cover this descriptor?
// (slabs narrower than 64 channels only up to 512 pixels: at 1024 pixels their idle lanes and 96-byte rows measured
// slower than the two-sweep kernels, 142 against 97 us on a 384-channel 32x32 layer at batch 128)
bool fused_ok(const GnArgs& a) { return g_dm_gn_fused && a.sw > 0 && a.HW <= (a.sw == 64 ? 1024 : 512); }
// slab width: the largest run of whole groups that is a multiple of 8 channels and at most 64 (0 if there is none)
int slab_width(int cg) {
  int l = cg;
  while (l % 8) l += cg;   // lcm(cg, 8)
  return l <= 64 ? 64 / l * l : 0;
}
// pixels per thread: the smallest of 1, 2, 4, 8 with at most 64 pixel lanes (128 beyond 512 pixels)
int fused_ppt(const GnArgs& a, bool bwd) {
  for (int ppt = 1; ppt < 8; ppt *= 2)
    if (dm_cdiv(a.HW, ppt) <= 64) {
      // Small slabs: a CU holds at most 32 waves, so with one pixel per thread it has 32 KB per tensor in flight, and
      // a workgroup lives for one memory round trip plus its reductions (~5 us) whatever its size: the kernel runs at
      // (bytes in flight) / (that latency) - 3 TB/s at 64 pixels. More pixels per thread in smaller workgroups put more
      // bytes in flight per wave (g_dm_gn_min_ppt; never fewer than 8 pixel lanes, one wave).
      while (ppt < g_dm_gn_min_ppt && ppt < 8 && dm_cdiv(a.HW, 2 * ppt) >= 8) ppt *= 2;
      return ppt;
    }
  return (bwd && g_dm_gn_bwd16 && a.HW > 512) ? 16 : 8;
}
int fused_threads(const GnArgs& a, bool bwd) {   // one thread per (pixel lane, 8-channel vector)
  const int lanes = dm_cdiv(a.HW, fused_ppt(a, bwd));
  return dm_cdiv(lanes * 8, 64) * 64;
}

template <bool BWD>
void launch_fused(const GnArgs& a, hipStream_t st) {
  const dim3 grid(dm_cdiv(a.C, a.sw), a.B), block(fused_threads(a, BWD));
  const int ppt = fused_ppt(a, BWD);
  const bool exact = a.sw == 64 && a.C % 64 == 0 && a.HW == ppt * (int)(block.x >> 3);
#define DM_GN_LAUNCH(P, T)                                                                                   \
  do {                                                                                                       \
    if (BWD) {                                                                                               \
      if (exact && a.act && a.dextra) hipLaunchKernelGGL((gn_bwd_fused_kernel<P, T, true, true, true>), grid, block, 0, st, a); \
      else if (exact && a.act) hipLaunchKernelGGL((gn_bwd_fused_kernel<P, T, true, false, true>), grid, block, 0, st, a); \
      else if (a.act && a.dextra) hipLaunchKernelGGL((gn_bwd_fused_kernel<P, T, true, true, false>), grid, block, 0, st, a);   \
      else if (a.act) hipLaunchKernelGGL((gn_bwd_fused_kernel<P, T, true, false, false>), grid, block, 0, st, a);   \
      else if (a.dextra) hipLaunchKernelGGL((gn_bwd_fused_kernel<P, T, false, true, false>), grid, block, 0, st, a); \
      else hipLaunchKernelGGL((gn_bwd_fused_kernel<P, T, false, false, false>), grid, block, 0, st, a);             \
    } else if (a.act) hipLaunchKernelGGL((gn_fwd_fused_kernel<P, T, true>), grid, block, 0, st, a);          \
    else hipLaunchKernelGGL((gn_fwd_fused_kernel<P, T, false>), grid, block, 0, st, a);                      \
  } while (0)
  if (ppt == 1) DM_GN_LAUNCH(1, 512);
  else if (ppt == 2) DM_GN_LAUNCH(2, 512);
  else if (ppt == 4) DM_GN_LAUNCH(4, 512);
  else if (a.HW <= 512) DM_GN_LAUNCH(8, 512);
  else if (BWD && g_dm_gn_bwd16) DM_GN_LAUNCH(16, 512);
  else DM_GN_LAUNCH(8, 1024);
#undef DM_GN_LAUNCH
}

int check_desc(const dm_gn_desc* d, const char* who) {
  DM_REQUIRE(d, DM_ERR_ARG, "%s: null descriptor", who);
  const int C = d->C1 + d->C2;
  DM_REQUIRE(d->B > 0 && d->HW > 0 && d->C1 > 0 && d->C2 >= 0 && d->G > 0, DM_ERR_SHAPE, "%s: empty tensor", who);
  DM_REQUIRE(d->C1 % 8 == 0 && d->C2 % 8 == 0, DM_ERR_SHAPE, "%s: channels (%d, %d) must be multiples of 8", who, d->C1, d->C2);
  DM_REQUIRE(C % d->G == 0, DM_ERR_SHAPE, "%s: %d channels not divisible into %d groups", who, C, d->G);
  DM_REQUIRE(C <= MAXC && d->G <= 256, DM_ERR_UNSUPPORTED, "%s: at most %d channels and 256 groups", who, MAXC);
  DM_REQUIRE(d->B <= 65535, DM_ERR_UNSUPPORTED, "%s: batch %d > 65535", who, d->B);
  DM_REQUIRE(!(d->has_ss && d->has_pre), DM_ERR_ARG, "%s: scale_shift and pre_add are exclusive", who);
  return DM_OK;
}

// pixels per workgroup: about 1024 workgroups (measured best: fewer, longer workgroups amortise the per-workgroup
// constants and reduction), at least 4 unrolled iterations per thread
int pick_chunk(int B, int HW, int C) {
  const int PL = NT / (C / 8) > 0 ? NT / (C / 8) : 1;
  int pch = 4 * PL;
  while ((long)B * dm_cdiv(HW, pch) > 1024 && pch < HW) pch *= 2;
  return pch;
}

void fill(GnArgs& a, const dm_gn_desc* d) {
  a.B = d->B; a.HW = d->HW; a.C1 = d->C1; a.C2 = d->C2; a.C = d->C1 + d->C2; a.G = d->G; a.cg = a.C / a.G;
  a.act = d->act; a.eps = d->eps;
  a.pch = pick_chunk(a.B, a.HW, a.C);
  a.sw = slab_width(a.cg);
}

}  // namespace

// per-device address of `scratch`, resolved on first use with that device current
constexpr int NOALLOC_MAX_DEVICES = 64;
static float* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;

static int scratch_for_current_device(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  DM_REQUIRE(e == hipSuccess, DM_ERR_LAUNCH, "no-allocation scratch: hipGetDevice failed: %s", hipGetErrorString(e));
  DM_REQUIRE(dev >= 0 && dev < NOALLOC_MAX_DEVICES, DM_ERR_UNSUPPORTED, "no-allocation scratch: device index %d >= %d", dev, NOALLOC_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    float* p = nullptr;
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    DM_REQUIRE(e == hipSuccess && p, DM_ERR_LAUNCH, "no-allocation scratch: hipGetSymbolAddress failed on device %d: %s", dev,
               hipGetErrorString(e));
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return DM_OK;
}

extern "C" int dm_noalloc_init(void) {
  float* p = nullptr;
  return scratch_for_current_device(&p);
}

extern "C" long dm_groupnorm_noalloc_max_ws_floats(void) { return NOALLOC_WS_FLOATS; }

extern "C" long dm_groupnorm_ws_floats(const dm_gn_desc* d) {
  return d ? 3L * d->B * (d->C1 + d->C2) : 0;
}

extern int g_dm_timing_skip;
extern int g_dm_timing_twice;

extern "C" int dm_groupnorm_act_fwd(const dm_gn_desc* d, const void* x1, const void* x2, const float* gamma,
                                    const float* beta, const float* scale_shift, const float* pre_add, void* y,
                                    float* mean, float* rstd, float* ws, void* stream) {
  int rc = check_desc(d, "dm_groupnorm_act_fwd");
  if (rc) return rc;
  if (g_dm_timing_skip & 16) return DM_OK;   // timing experiments only
  {
    static thread_local bool again = false;
    if ((g_dm_timing_twice & 16) && !again) {
      again = true;
      dm_groupnorm_act_fwd(d, x1, x2, gamma, beta, scale_shift, pre_add, y, mean, rstd, ws, stream);
      again = false;
    }
  }
  DM_REQUIRE(x1 && gamma && beta && y && mean && rstd && ws, DM_ERR_ARG, "dm_groupnorm_act_fwd: null argument");
  DM_REQUIRE((d->C2 == 0) == (x2 == nullptr), DM_ERR_ARG, "dm_groupnorm_act_fwd: x2 and C2 disagree");
  DM_REQUIRE((d->has_ss != 0) == (scale_shift != nullptr) && (d->has_pre != 0) == (pre_add != nullptr), DM_ERR_ARG,
             "dm_groupnorm_act_fwd: scale_shift/pre_add pointers disagree with the descriptor");
  DM_REQUIRE(((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)y | (uintptr_t)ws) % 16 == 0, DM_ERR_ARG, "dm_groupnorm_act_fwd: misaligned tensor");
  GnArgs a = {};
  a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.gamma = gamma; a.beta = beta; a.ss = scale_shift; a.pre = pre_add;
  a.y = (bf16_t*)y; a.mean = mean; a.rstd = rstd; a.ws = ws;
  fill(a, d);
  hipStream_t st = (hipStream_t)stream;
  if (fused_ok(a)) {
    launch_fused<false>(a, st);
    DM_CHECK_LAUNCH();
    return DM_OK;
  }
  dim3 grid(dm_cdiv(a.HW, a.pch), a.B), slabs(dm_cdiv(a.C, SL_C), a.B);
  hipLaunchKernelGGL(gn_stats_kernel, slabs, dim3(NT), 0, st, a);
  DM_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(NT), 0, st, a);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_groupnorm_act_fwd_noalloc(const dm_gn_desc* d, const void* x1, const void* x2, const float* gamma,
                                            const float* beta, const float* scale_shift, const float* pre_add, void* y,
                                            float* mean, float* rstd, void* stream) {
  int rc = check_desc(d, "dm_groupnorm_act_fwd_noalloc");
  if (rc) return rc;
  const long need = dm_groupnorm_ws_floats(d);
  DM_REQUIRE(need <= NOALLOC_WS_FLOATS, DM_ERR_UNSUPPORTED,
             "dm_groupnorm_act_fwd_noalloc: B * C = %ld needs %ld workspace floats, the built-in scratch holds %ld "
             "(use dm_groupnorm_act_fwd with a workspace of your own)", (long)d->B * (d->C1 + d->C2), need, NOALLOC_WS_FLOATS);
  float* ws = nullptr;
  rc = scratch_for_current_device(&ws);
  if (rc) return rc;
  // the per-channel sums are written before they are read in every call: nothing has to be cleared
  return dm_groupnorm_act_fwd(d, x1, x2, gamma, beta, scale_shift, pre_add, y, mean, rstd, ws, stream);
}

extern "C" int dm_groupnorm_act_bwd(const dm_gn_desc* d, const void* x1, const void* x2, const float* gamma,
                                    const float* beta, const float* scale_shift, const float* pre_add,
                                    const float* mean, const float* rstd, const void* dy, const void* dextra,
                                    const void* dextra2, void* dx1, void* dx2, float* dscale_shift, float* dpre_add,
                                    float* param_partials, int want_upstream_bias, float* ws, void* stream) {
  int rc = check_desc(d, "dm_groupnorm_act_bwd");
  if (rc) return rc;
  if (g_dm_timing_skip & 8) return DM_OK;    // timing experiments only
  {
    static thread_local bool again = false;
    if ((g_dm_timing_twice & 8) && !again) {
      again = true;
      dm_groupnorm_act_bwd(d, x1, x2, gamma, beta, scale_shift, pre_add, mean, rstd, dy, dextra, dextra2, dx1, dx2,
                           dscale_shift, dpre_add, param_partials, want_upstream_bias, ws, stream);
      again = false;
    }
  }
  DM_REQUIRE(x1 && gamma && beta && mean && rstd && dy && dx1 && param_partials && ws, DM_ERR_ARG,
             "dm_groupnorm_act_bwd: null argument");
  DM_REQUIRE((d->C2 == 0) == (x2 == nullptr) && (d->C2 == 0) == (dx2 == nullptr), DM_ERR_ARG, "dm_groupnorm_act_bwd: x2/dx2 and C2 disagree");
  DM_REQUIRE((d->has_ss != 0) == (scale_shift != nullptr) && (d->has_ss != 0) == (dscale_shift != nullptr), DM_ERR_ARG,
             "dm_groupnorm_act_bwd: scale_shift pointers disagree with the descriptor");
  DM_REQUIRE((d->has_pre != 0) == (pre_add != nullptr) && (d->has_pre != 0) == (dpre_add != nullptr), DM_ERR_ARG,
             "dm_groupnorm_act_bwd: pre_add pointers disagree with the descriptor");
  DM_REQUIRE(((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)dy | (uintptr_t)dextra | (uintptr_t)dextra2 | (uintptr_t)dx1 |
              (uintptr_t)dx2 | (uintptr_t)ws) % 16 == 0,
             DM_ERR_ARG, "dm_groupnorm_act_bwd: misaligned tensor");
  DM_REQUIRE(!dextra2 || dextra, DM_ERR_ARG, "dm_groupnorm_act_bwd: dextra2 without dextra (pass a single addend as dextra)");
  GnArgs a = {};
  a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.gamma = gamma; a.beta = beta; a.ss = scale_shift; a.pre = pre_add;
  a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
  a.dy = (const bf16_t*)dy; a.dextra = (const bf16_t*)dextra; a.dextra2 = (const bf16_t*)dextra2;
  a.dx1 = (bf16_t*)dx1; a.dx2 = (bf16_t*)dx2;
  DM_REQUIRE(!(want_upstream_bias && dextra), DM_ERR_ARG, "dm_groupnorm_act_bwd: the upstream bias gradient excludes dextra");
  a.dss = dscale_shift; a.dpre = dpre_add; a.pp = param_partials; a.want_dup = want_upstream_bias; a.ws = ws;
  fill(a, d);
  hipStream_t st = (hipStream_t)stream;
  if (fused_ok(a)) {
    launch_fused<true>(a, st);
    DM_CHECK_LAUNCH();
    return DM_OK;
  }
  dim3 grid(dm_cdiv(a.HW, a.pch), a.B), slabs(dm_cdiv(a.C, SL_C), a.B);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, slabs, dim3(NT), 0, st, a);
  DM_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_apply_kernel, grid, dim3(NT), 0, st, a);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_groupnorm_param_grads(const dm_gn_desc* d, const float* param_partials, float* dgamma, float* dbeta,
                                        float* dupstream_bias, dm_sum_item* deferred, void* stream) {
  int rc = check_desc(d, "dm_groupnorm_param_grads");
  if (rc) return rc;
  DM_REQUIRE(param_partials && dgamma && dbeta, DM_ERR_ARG, "dm_groupnorm_param_grads: null argument");
  const int C = d->C1 + d->C2;
  DmOrderedSum a = {param_partials, 3L * C, d->B, dupstream_bias ? 3 * C : 2 * C, C, 2 * C, dbeta, dgamma, dupstream_bias, nullptr};
  return dm_finish_ordered_sum(a, deferred, (hipStream_t)stream);
}
