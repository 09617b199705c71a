// Fused GroupNorm (+ per-sample scale/shift or pre-add) (+ SiLU), forward and backward, NHWC bf16.
// HBM-bound: one workgroup owns one (sample, channel slab); the slab is read from HBM once and re-read from L2
// for the second/third sweep. Channel slabs are whole groups and whole 8-channel (16-byte) vectors, so every
// access is a 16-byte piece of a pixel row and the channel axis may be split over two source tensors (x1 | x2).
#include "dm_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXCS = 1024;

struct GnArgs {
  const bf16_t* x1;
  const bf16_t* x2;
  const float* gamma;
  const float* beta;
  const float* ss;
  const float* pre;
  const float* mean_in;
  const float* rstd_in;
  const bf16_t* dy;
  const bf16_t* dextra;
  bf16_t* y;     // fwd output
  bf16_t* dx1;
  bf16_t* dx2;
  float* mean;
  float* rstd;
  float* dgamma;
  float* dbeta;
  float* dss;
  float* dpre;
  int B, HW, C1, C2, C, G, cg, CS, act;
  float eps;
};

__device__ __forceinline__ bf16x8_t load_x(const GnArgs& a, int b, int p, int c) {
  if (c < a.C1) return *(const bf16x8_t*)(a.x1 + ((size_t)b * a.HW + p) * a.C1 + c);
  return *(const bf16x8_t*)(a.x2 + ((size_t)b * a.HW + p) * a.C2 + (c - a.C1));
}

__global__ __launch_bounds__(NT) void gn_fwd_kernel(GnArgs a) {
  __shared__ float s_a[MAXCS];
  __shared__ float s_b[MAXCS];
  __shared__ float s_mean[MAXCS];
  __shared__ float s_rstd[MAXCS];
  const int b = blockIdx.y, c0 = blockIdx.x * a.CS;
  const int tid = threadIdx.x;
  const int VP = a.CS >> 3, PL = NT / VP;
  const bool active = tid < PL * VP;
  const int v = tid % VP, pl = tid / VP;
  const int cl = 8 * v, c = c0 + cl;
  for (int i = tid; i < a.CS; i += NT) { s_a[i] = 0.f; s_b[i] = 0.f; }
  __syncthreads();

  float pre[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) pre[j] = (a.pre && active) ? a.pre[(size_t)b * a.C + c + j] : 0.f;

  // sweep 1: per-channel sums
  if (active) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = pl; p < a.HW; p += PL) {
      float f[8];
      dm_unpack8(load_x(a, b, p, c), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j] + pre[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&s_a[cl + j], s[j]);
  }
  __syncthreads();
  const int ngs = a.CS / a.cg;
  const float inv_n = 1.0f / (float)(a.cg * a.HW);
  for (int g = tid; g < ngs; g += NT) {
    float t = 0.f;
    for (int j = 0; j < a.cg; ++j) t += s_a[g * a.cg + j];
    s_mean[g] = t * inv_n;
  }
  __syncthreads();
  float mu[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) mu[j] = active ? s_mean[(cl + j) / a.cg] : 0.f;
  // sweep 2: centred second moment
  if (active) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = pl; p < a.HW; p += PL) {
      float f[8];
      dm_unpack8(load_x(a, b, p, c), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = f[j] + pre[j] - mu[j]; s[j] += d * d; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&s_b[cl + j], s[j]);
  }
  __syncthreads();
  for (int gl = tid; gl < ngs; gl += NT) {
    float t = 0.f;
    for (int j = 0; j < a.cg; ++j) t += s_b[gl * a.cg + j];
    float r = rsqrtf(t * inv_n + a.eps);
    s_rstd[gl] = r;
    const int g = c0 / a.cg + gl;
    a.mean[(size_t)b * a.G + g] = s_mean[gl];
    a.rstd[(size_t)b * a.G + g] = r;
  }
  __syncthreads();
  if (!active) return;
  // sweep 3: normalise, affine, scale/shift, activation
  float ka[8], kb[8];  // y = act(ka * xin + kb)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float r = s_rstd[(cl + j) / a.cg];
    float ga = a.gamma[c + j], be = a.beta[c + j];
    float sc = 1.f, sh = 0.f;
    if (a.ss) { sc = 1.f + a.ss[(size_t)b * 2 * a.C + c + j]; sh = a.ss[(size_t)b * 2 * a.C + a.C + c + j]; }
    // u = sc * (ga * (x - mu) * r + be) + sh
    ka[j] = sc * ga * r;
    kb[j] = sc * (be - ga * r * mu[j]) + sh + ka[j] * pre[j];
  }
  for (int p = pl; p < a.HW; p += PL) {
    float f[8];
    dm_unpack8(load_x(a, b, p, c), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float u = ka[j] * f[j] + kb[j];
      f[j] = a.act ? dm_silu(u) : u;
    }
    *(bf16x8_t*)(a.y + ((size_t)b * a.HW + p) * a.C + c) = dm_pack8(f);
  }
}

__global__ __launch_bounds__(NT) void gn_bwd_kernel(GnArgs a) {
  __shared__ float s_a[MAXCS];
  __shared__ float s_b[MAXCS];
  __shared__ float s_c[MAXCS];
  __shared__ float s_A[MAXCS];
  __shared__ float s_B[MAXCS];
  const int b = blockIdx.y, c0 = blockIdx.x * a.CS;
  const int tid = threadIdx.x;
  const int VP = a.CS >> 3, PL = NT / VP;
  const bool active = tid < PL * VP;
  const int v = tid % VP, pl = tid / VP;
  const int cl = 8 * v, c = c0 + cl;
  for (int i = tid; i < a.CS; i += NT) { s_a[i] = 0.f; s_b[i] = 0.f; s_c[i] = 0.f; }
  __syncthreads();

  // per-channel constants: xhat = (x + pre - mu) * r ; u = sc * (ga * xhat + be) + sh
  float pre[8], mu[8], r[8], ga[8], be[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (active) {
      const int g = (c + j) / a.cg;
      pre[j] = a.pre ? a.pre[(size_t)b * a.C + c + j] : 0.f;
      mu[j] = a.mean_in[(size_t)b * a.G + g];
      r[j] = a.rstd_in[(size_t)b * a.G + g];
      ga[j] = a.gamma[c + j];
      be[j] = a.beta[c + j];
      sc[j] = a.ss ? 1.f + a.ss[(size_t)b * 2 * a.C + c + j] : 1.f;
      sh[j] = a.ss ? a.ss[(size_t)b * 2 * a.C + a.C + c + j] : 0.f;
    } else {
      pre[j] = mu[j] = r[j] = ga[j] = be[j] = sh[j] = 0.f; sc[j] = 1.f;
    }
  }
  // sweep 1: a_c = sum du, b_c = sum du * xhat
  if (active) {
    float sa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = pl; p < a.HW; p += PL) {
      float f[8], d[8];
      dm_unpack8(load_x(a, b, p, c), f);
      dm_unpack8(*(const bf16x8_t*)(a.dy + ((size_t)b * a.HW + p) * a.C + c), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xh = (f[j] + pre[j] - mu[j]) * r[j];
        float u = sc[j] * (ga[j] * xh + be[j]) + sh[j];
        float du = a.act ? d[j] * dm_silu_grad(u) : d[j];
        sa[j] += du;
        sb[j] += du * xh;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { atomicAdd(&s_a[cl + j], sa[j]); atomicAdd(&s_b[cl + j], sb[j]); }
  }
  __syncthreads();
  const int ngs = a.CS / a.cg;
  for (int gl = tid; gl < ngs; gl += NT) {
    float A = 0.f, Bv = 0.f;
    for (int j = 0; j < a.cg; ++j) {
      int ch = c0 + gl * a.cg + j;
      float k = a.gamma[ch] * (a.ss ? 1.f + a.ss[(size_t)b * 2 * a.C + ch] : 1.f);
      A += k * s_a[gl * a.cg + j];
      Bv += k * s_b[gl * a.cg + j];
    }
    const float inv_n = 1.0f / (float)(a.cg * a.HW);
    s_A[gl] = A * inv_n;
    s_B[gl] = Bv * inv_n;
  }
  // per-channel parameter-side gradients of this sample
  for (int i = tid; i < a.CS; i += NT) {
    const int ch = c0 + i;
    const float av = s_a[i], bv = s_b[i];
    const float scv = a.ss ? 1.f + a.ss[(size_t)b * 2 * a.C + ch] : 1.f;
    atomicAdd(a.dbeta + ch, scv * av);
    atomicAdd(a.dgamma + ch, scv * bv);
    if (a.dss) {
      a.dss[(size_t)b * 2 * a.C + ch] = a.gamma[ch] * bv + a.beta[ch] * av;
      a.dss[(size_t)b * 2 * a.C + a.C + ch] = av;
    }
  }
  __syncthreads();
  // sweep 2: dx
  if (active) {
    float Ag[8], Bg[8], sp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) { Ag[j] = s_A[(cl + j) / a.cg]; Bg[j] = s_B[(cl + j) / a.cg]; }
    for (int p = pl; p < a.HW; p += PL) {
      float f[8], d[8], e[8];
      dm_unpack8(load_x(a, b, p, c), f);
      const size_t o = ((size_t)b * a.HW + p) * a.C + c;
      dm_unpack8(*(const bf16x8_t*)(a.dy + o), d);
      if (a.dextra) dm_unpack8(*(const bf16x8_t*)(a.dextra + o), e);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xh = (f[j] + pre[j] - mu[j]) * r[j];
        float u = sc[j] * (ga[j] * xh + be[j]) + sh[j];
        float du = a.act ? d[j] * dm_silu_grad(u) : d[j];
        float dxv = r[j] * (sc[j] * ga[j] * du - Ag[j] - xh * Bg[j]);
        sp[j] += dxv;
        f[j] = a.dextra ? dxv + e[j] : dxv;
      }
      bf16x8_t ov = dm_pack8(f);
      if (c < a.C1) *(bf16x8_t*)(a.dx1 + ((size_t)b * a.HW + p) * a.C1 + c) = ov;
      else *(bf16x8_t*)(a.dx2 + ((size_t)b * a.HW + p) * a.C2 + (c - a.C1)) = ov;
    }
    if (a.dpre) {
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&s_c[cl + j], sp[j]);
    }
  }
  if (a.dpre) {
    __syncthreads();
    for (int i = tid; i < a.CS; i += NT) a.dpre[(size_t)b * a.C + c0 + i] = s_c[i];
  }
}

int gcd_i(int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; }

// channels per slab: whole groups, whole 8-channel vectors, enough workgroups to fill the chip
int pick_slab(int B, int C, int cg) {
  const int unit = cg / gcd_i(cg, 8) * 8;
  int best = 0;
  for (int nslab = 1; nslab <= C / unit; ++nslab) {
    if (C % nslab) continue;
    const int cs = C / nslab;
    if (cs % unit || cs > MAXCS) continue;
    if (cs < 32 && best) break;  // keep at least 64-byte pieces of a pixel row per slab
    best = cs;
    if ((long)B * nslab >= 768) break;
  }
  return best;
}

int check_desc(const dm_gn_desc* d, const char* who) {
  DM_REQUIRE(d, DM_ERR_ARG, "%s: null descriptor", who);
  const int C = d->C1 + d->C2;
  DM_REQUIRE(d->B > 0 && d->HW > 0 && d->C1 > 0 && d->C2 >= 0 && d->G > 0, DM_ERR_SHAPE, "%s: empty tensor", who);
  DM_REQUIRE(d->C1 % 8 == 0 && d->C2 % 8 == 0, DM_ERR_SHAPE, "%s: channels (%d, %d) must be multiples of 8", who, d->C1, d->C2);
  DM_REQUIRE(C % d->G == 0, DM_ERR_SHAPE, "%s: %d channels not divisible into %d groups", who, C, d->G);
  const int cg = C / d->G;
  const int unit = cg / gcd_i(cg, 8) * 8;
  DM_REQUIRE(C % unit == 0 && pick_slab(d->B, C, cg) > 0, DM_ERR_SHAPE, "%s: group size %d cannot be tiled by 16-byte vectors", who, cg);
  DM_REQUIRE(!(d->has_ss && d->has_pre), DM_ERR_ARG, "%s: scale_shift and pre_add are exclusive", who);
  return DM_OK;
}

}  // namespace

extern "C" int dm_groupnorm_act_fwd(const dm_gn_desc* d, const void* x1, const void* x2, const float* gamma,
                                    const float* beta, const float* scale_shift, const float* pre_add, void* y,
                                    float* mean, float* rstd, void* stream) {
  int rc = check_desc(d, "dm_groupnorm_act_fwd");
  if (rc) return rc;
  DM_REQUIRE(x1 && gamma && beta && y && mean && rstd, DM_ERR_ARG, "dm_groupnorm_act_fwd: null argument");
  DM_REQUIRE((d->C2 == 0) == (x2 == nullptr), DM_ERR_ARG, "dm_groupnorm_act_fwd: x2 and C2 disagree");
  DM_REQUIRE((d->has_ss != 0) == (scale_shift != nullptr) && (d->has_pre != 0) == (pre_add != nullptr), DM_ERR_ARG,
             "dm_groupnorm_act_fwd: scale_shift/pre_add pointers disagree with the descriptor");
  DM_REQUIRE(((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)y) % 16 == 0, DM_ERR_ARG, "dm_groupnorm_act_fwd: misaligned tensor");
  GnArgs a = {};
  a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.gamma = gamma; a.beta = beta; a.ss = scale_shift; a.pre = pre_add;
  a.y = (bf16_t*)y; a.mean = mean; a.rstd = rstd;
  a.B = d->B; a.HW = d->HW; a.C1 = d->C1; a.C2 = d->C2; a.C = d->C1 + d->C2; a.G = d->G; a.cg = a.C / a.G;
  a.CS = pick_slab(a.B, a.C, a.cg); a.act = d->act; a.eps = d->eps;
  hipLaunchKernelGGL(gn_fwd_kernel, dim3(a.C / a.CS, a.B), dim3(NT), 0, (hipStream_t)stream, a);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_groupnorm_act_bwd(const dm_gn_desc* d, const void* x1, const void* x2, const float* gamma,
                                    const float* beta, const float* scale_shift, const float* pre_add,
                                    const float* mean, const float* rstd, const void* dy, const void* dextra,
                                    void* dx1, void* dx2, float* dgamma, float* dbeta,
                                    float* dscale_shift, float* dpre_add, void* stream) {
  int rc = check_desc(d, "dm_groupnorm_act_bwd");
  if (rc) return rc;
  DM_REQUIRE(x1 && gamma && beta && mean && rstd && dy && dx1 && dgamma && dbeta, DM_ERR_ARG,
             "dm_groupnorm_act_bwd: null argument");
  DM_REQUIRE((d->C2 == 0) == (x2 == nullptr) && (d->C2 == 0) == (dx2 == nullptr), DM_ERR_ARG, "dm_groupnorm_act_bwd: x2/dx2 and C2 disagree");
  DM_REQUIRE((d->has_ss != 0) == (scale_shift != nullptr) && (d->has_ss != 0) == (dscale_shift != nullptr), DM_ERR_ARG,
             "dm_groupnorm_act_bwd: scale_shift pointers disagree with the descriptor");
  DM_REQUIRE((d->has_pre != 0) == (pre_add != nullptr) && (d->has_pre != 0) == (dpre_add != nullptr), DM_ERR_ARG,
             "dm_groupnorm_act_bwd: pre_add pointers disagree with the descriptor");
  DM_REQUIRE(((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)dy | (uintptr_t)dextra | (uintptr_t)dx1 | (uintptr_t)dx2) % 16 == 0,
             DM_ERR_ARG, "dm_groupnorm_act_bwd: misaligned tensor");
  GnArgs a = {};
  a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.gamma = gamma; a.beta = beta; a.ss = scale_shift; a.pre = pre_add;
  a.mean_in = mean; a.rstd_in = rstd; a.dy = (const bf16_t*)dy; a.dextra = (const bf16_t*)dextra;
  a.dx1 = (bf16_t*)dx1; a.dx2 = (bf16_t*)dx2; a.dgamma = dgamma; a.dbeta = dbeta;
  a.dss = dscale_shift; a.dpre = dpre_add;
  a.B = d->B; a.HW = d->HW; a.C1 = d->C1; a.C2 = d->C2; a.C = d->C1 + d->C2; a.G = d->G; a.cg = a.C / a.G;
  a.CS = pick_slab(a.B, a.C, a.cg); a.act = d->act; a.eps = d->eps;
  hipLaunchKernelGGL(gn_bwd_kernel, dim3(a.C / a.CS, a.B), dim3(NT), 0, (hipStream_t)stream, a);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
