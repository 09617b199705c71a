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
//                the parameter-side gradients (dgamma/dbeta atomics, dscale_shift, dpre_add)
// The second sweep of each pair re-reads a tensor that was just read, which the 256 MiB Infinity Cache serves.
// The workspace is pure scratch (written before it is read in every call), so one buffer serves every call.
// The channel axis may be the concatenation of two tensors (x1 | x2).
#include "dm_common.h"

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
  bf16_t* y;
  bf16_t* dx1;
  bf16_t* dx2;
  float* mean;
  float* rstd;
  float* dgamma;
  float* dbeta;
  float* dss;
  float* dpre;
  float* dup;   // bias gradient of the convolution that produced x (sum of dx over samples and pixels), or null
  float* ws;
  int B, HW, C1, C2, C, G, cg, act, pch;
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
      atomicAdd(a.dbeta + c, sc * av);
      atomicAdd(a.dgamma + c, sc * bv);
      if (a.dss) {
        a.dss[(size_t)b * 2 * a.C + c] = a.gamma[c] * bv + a.beta[c] * av;
        a.dss[(size_t)b * 2 * a.C + a.C + c] = av;
      }
      if (a.dpre || a.dup) {
        const int g = c / a.cg;
        // sum over pixels of dx = r * (k * a_c - HW * A_g - B_g * sum xhat_c)
        const float dxsum = s_xa[c] * (sc * a.gamma[c] * av - (float)a.HW * s_A[g] - s_B[g] * wx[c]);
        if (a.dpre) a.dpre[(size_t)b * a.C + c] = dxsum;
        if (a.dup) atomicAdd(a.dup + c, dxsum);
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
    float f[8], e[8];
    if (a.dextra) dm_unpack8(*(const bf16x8_t*)(a.dextra + o), e);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xv = (float)xr[j];
      const float xh = xa[j] * xv + xb[j];
      const float du = a.act ? (float)dr[j] * dm_silu_grad(ua[j] * xv + ub[j]) : (float)dr[j];
      // k * r = ua (u = ua * x + ub with ua = sc * gamma * r)
      const float dxv = ua[j] * du - xa[j] * (Ag[j] + xh * Bg[j]);
      f[j] = a.dextra ? dxv + e[j] : dxv;
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
}

}  // namespace

extern "C" long dm_groupnorm_ws_floats(const dm_gn_desc* d) {
  return d ? 3L * d->B * (d->C1 + d->C2) : 0;
}

extern "C" int dm_groupnorm_act_fwd(const dm_gn_desc* d, const void* x1, const void* x2, const float* gamma,
                                    const float* beta, const float* scale_shift, const float* pre_add, void* y,
                                    float* mean, float* rstd, float* ws, void* stream) {
  int rc = check_desc(d, "dm_groupnorm_act_fwd");
  if (rc) return rc;
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
  dim3 grid(dm_cdiv(a.HW, a.pch), a.B), slabs(dm_cdiv(a.C, SL_C), a.B);
  hipLaunchKernelGGL(gn_stats_kernel, slabs, dim3(NT), 0, st, a);
  DM_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(NT), 0, st, a);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_groupnorm_act_bwd(const dm_gn_desc* d, const void* x1, const void* x2, const float* gamma,
                                    const float* beta, const float* scale_shift, const float* pre_add,
                                    const float* mean, const float* rstd, const void* dy, const void* dextra,
                                    void* dx1, void* dx2, float* dgamma, float* dbeta, float* dscale_shift,
                                    float* dpre_add, float* dupstream_bias, float* ws, void* stream) {
  int rc = check_desc(d, "dm_groupnorm_act_bwd");
  if (rc) return rc;
  DM_REQUIRE(x1 && gamma && beta && mean && rstd && dy && dx1 && dgamma && dbeta && ws, DM_ERR_ARG,
             "dm_groupnorm_act_bwd: null argument");
  DM_REQUIRE((d->C2 == 0) == (x2 == nullptr) && (d->C2 == 0) == (dx2 == nullptr), DM_ERR_ARG, "dm_groupnorm_act_bwd: x2/dx2 and C2 disagree");
  DM_REQUIRE((d->has_ss != 0) == (scale_shift != nullptr) && (d->has_ss != 0) == (dscale_shift != nullptr), DM_ERR_ARG,
             "dm_groupnorm_act_bwd: scale_shift pointers disagree with the descriptor");
  DM_REQUIRE((d->has_pre != 0) == (pre_add != nullptr) && (d->has_pre != 0) == (dpre_add != nullptr), DM_ERR_ARG,
             "dm_groupnorm_act_bwd: pre_add pointers disagree with the descriptor");
  DM_REQUIRE(((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)dy | (uintptr_t)dextra | (uintptr_t)dx1 | (uintptr_t)dx2 | (uintptr_t)ws) % 16 == 0,
             DM_ERR_ARG, "dm_groupnorm_act_bwd: misaligned tensor");
  GnArgs a = {};
  a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.gamma = gamma; a.beta = beta; a.ss = scale_shift; a.pre = pre_add;
  a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
  a.dy = (const bf16_t*)dy; a.dextra = (const bf16_t*)dextra;
  a.dx1 = (bf16_t*)dx1; a.dx2 = (bf16_t*)dx2; a.dgamma = dgamma; a.dbeta = dbeta;
  DM_REQUIRE(!(dupstream_bias && dextra), DM_ERR_ARG, "dm_groupnorm_act_bwd: dupstream_bias excludes dextra");
  a.dss = dscale_shift; a.dpre = dpre_add; a.dup = dupstream_bias; a.ws = ws;
  fill(a, d);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(dm_cdiv(a.HW, a.pch), a.B), slabs(dm_cdiv(a.C, SL_C), a.B);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, slabs, dim3(NT), 0, st, a);
  DM_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_apply_kernel, grid, dim3(NT), 0, st, a);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
