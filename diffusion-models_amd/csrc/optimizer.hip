// Optimizer tail on flat fp32 arenas: global gradient norm, and one fused pass for
// clip -> AdamW (decoupled weight decay, bias correction) -> EMA -> bf16 weight copy.
// HBM-bound: 4 B/param for the norm, 28 B/param (+8 EMA, +2 bf16, +4 gradient zeroing) for the update.
#include "dm_common.h"

namespace {

constexpr int NT = 256;
constexpr int NPART = 1024;

__global__ __launch_bounds__(NT) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partials) {
  __shared__ float red[NT / 64];
  float s = 0.f;
  const long n4 = n >> 2;
  const f32x4_t* g4 = (const f32x4_t*)g;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
    const f32x4_t v = g4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0) for (long i = (n4 << 2) + threadIdx.x; i < n; i += NT) s += g[i] * g[i];
  s = dm_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(NT) void norm_final_kernel(const float* __restrict__ partials, int np, float grad_scale,
                                                       float* __restrict__ out) {
  __shared__ double red[NT];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += NT) s += (double)partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = NT / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = fabsf(grad_scale) * (float)sqrt(red[0]);
}

template <bool ZERO>
__global__ __launch_bounds__(NT) void adamw_ema_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ ema, bf16_t* __restrict__ wb,
                                                       const float* __restrict__ norm, float max_norm, long n, float lr,
                                                       float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                       float ema_decay, const float* __restrict__ hyper, float grad_scale) {
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; ema_decay = hyper[3]; }
  float clip = 1.f;
  if (max_norm > 0.f && norm) {
    clip = max_norm / (norm[0] + 1e-6f);
    clip = clip < 1.f ? clip : 1.f;
  }
  const float step_size = lr / bc1;
  const float sqrt_bc2 = sqrtf(bc2);
  const float omd = 1.f - ema_decay;
  // (4-byte accesses on purpose: with five arenas read and five written per element, 16-byte accesses measured
  // slower - 490-506 against 443-449 us for 55.9 M parameters, 4.2-4.3 against 4.7-4.8 TB/s)
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const float gi = g[i] * grad_scale * clip;
    if (ZERO) g[i] = 0.f;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (ema) { const float e = ema[i]; ema[i] = e - omd * (e - pi); }
    if (wb) wb[i] = (bf16_t)pi;
  }
}

// advances the step counter and derives the step-dependent scalars on the device, so a captured HIP graph of a
// whole training step can be replayed: hyper = {lr, 1 - b1^step, 1 - b2^step, ema_decay}
// Multiplier of the base learning rate at optimizer step k (0-based), the rules of the reference's lr_scheduler
// (train.py:281-291: transformers.get_scheduler -> torch LambdaLR; formulas of transformers/optimization.py).
// p0, p1: num_cycles (cosine kinds), power and lr_end / lr_init (polynomial), timescale (inverse_sqrt).
__host__ __device__ inline double lr_multiplier(int kind, long long k, long long warmup, long long total, double p0,
                                                double p1) {
  const double PI = 3.14159265358979323846;
  if (kind == DM_LR_CONSTANT) return 1.0;
  if (k < warmup) return (double)k / (double)(warmup > 1 ? warmup : 1);
  const long long span = total - warmup > 1 ? total - warmup : 1;
  switch (kind) {
    case DM_LR_LINEAR: {
      const double f = (double)(total - k) / (double)span;
      return f > 0.0 ? f : 0.0;
    }
    case DM_LR_COSINE: {
      const double f = 0.5 * (1.0 + cos(PI * p0 * 2.0 * ((double)(k - warmup) / (double)span)));
      return f > 0.0 ? f : 0.0;
    }
    case DM_LR_COSINE_RESTARTS: {
      const double progress = (double)(k - warmup) / (double)span;
      if (progress >= 1.0) return 0.0;
      const double f = 0.5 * (1.0 + cos(PI * fmod(p0 * progress, 1.0)));
      return f > 0.0 ? f : 0.0;
    }
    case DM_LR_POLYNOMIAL: {   // p1 = lr_end / lr_init
      if (k > total) return p1;
      const double remaining = 1.0 - (double)(k - warmup) / (double)(total - warmup);
      return (1.0 - p1) * pow(remaining, p0) + p1;
    }
    case DM_LR_INVERSE_SQRT:   // p0 = timescale
      return 1.0 / sqrt(((double)k + (p0 - (double)warmup)) / p0);
    default:                   // DM_LR_CONSTANT_WITH_WARMUP
      return 1.0;
  }
}

__global__ void optimizer_tick_kernel(long long* step, float* hyper, float lr, float b1, float b2, float ema_inv_gamma,
                                      float ema_power, float ema_max_decay, int lr_kind, long long lr_warmup,
                                      long long lr_total, double lr_p0, double lr_p1) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long s = step[0] + 1;
  step[0] = s;
  hyper[0] = (float)((double)lr * lr_multiplier(lr_kind, s - 1, lr_warmup, lr_total, lr_p0, lr_p1));
  hyper[1] = (float)(1.0 - pow((double)b1, (double)s));
  hyper[2] = (float)(1.0 - pow((double)b2, (double)s));
  // warm-up EMA: decay = min(1 - (1 + (s-1)/inv_gamma)^-power, max), 0 on the first step
  double d = 0.0;
  if (s > 1) {
    d = 1.0 - pow(1.0 + (double)(s - 1) / (double)ema_inv_gamma, -(double)ema_power);
    if (d > (double)ema_max_decay) d = (double)ema_max_decay;
    if (d < 0.0) d = 0.0;
  }
  hyper[3] = (float)d;
}

}  // namespace

extern "C" int dm_optimizer_tick(long long* step, float* hyper, float lr, float b1, float b2, float ema_inv_gamma,
                                 float ema_power, float ema_max_decay, void* stream) {
  return dm_optimizer_tick_lr(step, hyper, lr, DM_LR_CONSTANT, 0, 0, 0.0, 0.0, b1, b2, ema_inv_gamma, ema_power,
                              ema_max_decay, stream);
}

extern "C" int dm_optimizer_tick_lr(long long* step, float* hyper, float base_lr, int kind, long long warmup,
                                    long long total, double p0, double p1, float b1, float b2, float ema_inv_gamma,
                                    float ema_power, float ema_max_decay, void* stream) {
  DM_REQUIRE(step && hyper, DM_ERR_ARG, "dm_optimizer_tick_lr: null argument");
  DM_REQUIRE(kind >= DM_LR_CONSTANT && kind <= DM_LR_INVERSE_SQRT, DM_ERR_ARG, "dm_optimizer_tick_lr: unknown schedule %d", kind);
  DM_REQUIRE(warmup >= 0 && total >= 0, DM_ERR_ARG, "dm_optimizer_tick_lr: negative step counts");
  DM_REQUIRE(kind != DM_LR_POLYNOMIAL || total > warmup, DM_ERR_ARG,
             "dm_optimizer_tick_lr: the polynomial schedule needs total > warmup steps");
  DM_REQUIRE(kind != DM_LR_INVERSE_SQRT || p0 > 0.0, DM_ERR_ARG, "dm_optimizer_tick_lr: the inverse-sqrt timescale must be positive");
  hipLaunchKernelGGL(optimizer_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, hyper, base_lr, b1, b2,
                     ema_inv_gamma, ema_power, ema_max_decay, kind, warmup, total, p0, p1);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" double dm_lr_multiplier(int kind, long long k, long long warmup, long long total, double p0, double p1) {
  return lr_multiplier(kind, k, warmup, total, p0, p1);
}

extern "C" int dm_sumsq_partials(void) { return NPART; }

extern "C" int dm_grad_norm(const float* g, long n, float grad_scale, float* partials, float* norm_out, void* stream) {
  DM_REQUIRE(g && partials && norm_out && n > 0, DM_ERR_ARG, "dm_grad_norm: bad argument");
  DM_REQUIRE((uintptr_t)g % 16 == 0, DM_ERR_ARG, "dm_grad_norm: gradient arena must be 16-byte aligned");
  long gl = ((n >> 2) + NT - 1) / NT;
  int grid = (int)(gl > NPART ? NPART : (gl < 1 ? 1 : gl));
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, g, n, partials);
  DM_CHECK_LAUNCH();
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partials, grid, grad_scale, norm_out);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_adamw_ema_step(float* p, float* g, float* m, float* v, float* ema, void* w_bf16, const float* norm,
                                 float max_norm, long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                                 float bc2, float ema_decay, const float* hyper, float grad_scale, int zero_grad,
                                 void* stream) {
  DM_REQUIRE(p && g && m && v && n > 0, DM_ERR_ARG, "dm_adamw_ema_step: bad argument");
  DM_REQUIRE(!(max_norm > 0.f) || norm, DM_ERR_ARG, "dm_adamw_ema_step: clipping needs the norm");
  long gl = (n + NT - 1) / NT;
  int grid = (int)(gl > 4096 ? 4096 : gl);
  if (zero_grad)
    hipLaunchKernelGGL(adamw_ema_kernel<true>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, ema, (bf16_t*)w_bf16,
                       norm, max_norm, n, lr, b1, b2, eps, wd, bc1, bc2, ema_decay, hyper, grad_scale);
  else
    hipLaunchKernelGGL(adamw_ema_kernel<false>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, ema, (bf16_t*)w_bf16,
                       norm, max_norm, n, lr, b1, b2, eps, wd, bc1, bc2, ema_decay, hyper, grad_scale);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
