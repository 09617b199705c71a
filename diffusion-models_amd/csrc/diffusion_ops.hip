// Diffusion-process elementwise kernels (fp32, HBM-bound): forward noising, MSE loss and its gradient,
// the per-step sampler update and the classifier-free-guidance combine.
// Every product and sum must be rounded separately: the reference evaluates these expressions as separate fp32
// tensor ops, and the sampler update is iterated up to 1000 times. HIP's __fmul_rn/__fadd_rn are plain
// operators that the compiler would still contract into FMAs, so this file is built with -ffp-contract=off.
#include "dm_common.h"

namespace {

constexpr int NT = 256;
constexpr int NPART = 1024;

__global__ void diffuse_kernel(const float* __restrict__ x, const float* __restrict__ eps, const int64_t* __restrict__ t,
                               const double* __restrict__ tab64, const float* __restrict__ tab32, float* __restrict__ xt,
                               long n, int mode) {
  const int b = blockIdx.y;
  float a, s;
  if (mode == 0) {
    const double asq = tab64[t[b]];
    a = (float)sqrt(asq);
    s = (float)sqrt(1.0 - asq);
  } else {
    s = tab32[t[b]];
    a = __fsub_rn(1.0f, s);
  }
  const float* xb = x + (size_t)b * n;
  const float* eb = eps + (size_t)b * n;
  float* ob = xt + (size_t)b * n;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT)
    ob[i] = __fadd_rn(__fmul_rn(a, xb[i]), __fmul_rn(s, eb[i]));
}

__device__ __forceinline__ float target_of(const float* x, const float* eps, long i, int mode) {
  return mode == 0 ? eps[i] : __fsub_rn(x[i], eps[i]);
}

__global__ __launch_bounds__(NT) void mse_partial_kernel(const float* __restrict__ pred, const float* __restrict__ x,
                                                         const float* __restrict__ eps, float* __restrict__ partials,
                                                         long total, int mode) {
  __shared__ float red[NT / 64];
  float s = 0.f;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const float d = pred[i] - target_of(x, eps, i, mode);
    s += d * d;
  }
  s = dm_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(NT) void mse_final_kernel(const float* __restrict__ partials, int np, float* __restrict__ loss,
                                                       double inv_total) {
  __shared__ double red[NT];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += NT) s += (double)partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = NT / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] * inv_total);
}

__global__ void mse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ x, const float* __restrict__ eps,
                               const float* __restrict__ gout, float* __restrict__ dpred, long total, int mode, float k) {
  const float g = gout[0] * k;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT)
    dpred[i] = g * (pred[i] - target_of(x, eps, i, mode));
}

// one element of a sampler update (shared by the by-value and the device-table kernels: identical arithmetic)
__device__ __forceinline__ float sampler_update(float zi, float p, const float* noise, long i, float c0, float c1, float c2,
                                                float c3, int mode, int clip, int last) {
  float r;
  if (mode == 0) {
    r = __fmul_rn(c0, __fsub_rn(zi, __fmul_rn(c1, p)));
    if (noise) r = __fadd_rn(r, __fmul_rn(c2, noise[i]));
  } else if (mode == 1) {
    float x0 = __fdiv_rn(__fsub_rn(zi, __fmul_rn(c1, p)), c0);
    if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    r = last ? x0 : __fadd_rn(__fmul_rn(c2, x0), __fmul_rn(c3, p));
  } else {
    r = __fsub_rn(zi, __fmul_rn(c0, p));
  }
  return r;
}

// The update of loop step i = step[0], with the step's four scalars read from a device table: a captured HIP graph
// can then replay the same launch for every step of a sampling loop. The step's noise is ignored on the last step
// (the reference draws none there).
__global__ void sampler_step_dev_kernel(float* __restrict__ z, const float* __restrict__ pred,
                                        const float* __restrict__ noise, const float* __restrict__ coefs,
                                        const long long* __restrict__ step, int n_steps, int mode, int clip, long total) {
  const long long s = step[0];
  if (s < 0 || s >= n_steps) return;   // replayed more often than the loop has steps: leave z alone
  const float c0 = coefs[4 * s], c1 = coefs[4 * s + 1], c2 = coefs[4 * s + 2], c3 = coefs[4 * s + 3];
  const int last = s == n_steps - 1;
  const float* nz = last ? nullptr : noise;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT)
    z[i] = sampler_update(z[i], pred[i], nz, i, c0, c1, c2, c3, mode, clip, last);
}
// step[0] += 1 and t_out[b] = t_table[step] for the next UNet forward (one workgroup; runs after the update kernel)
__global__ void sampler_tick_kernel(long long* __restrict__ step, const int64_t* __restrict__ t_table,
                                    int64_t* __restrict__ t_out, int n_steps, int B) {
  const long long s = step[0] + 1;
  const int64_t t = t_table[s < n_steps ? s : n_steps - 1];
  for (int b = threadIdx.x; b < B; b += blockDim.x) t_out[b] = t;
  __syncthreads();
  if (threadIdx.x == 0) step[0] = s;
}

__global__ void sampler_step_kernel(float* __restrict__ z, const float* __restrict__ pred, const float* __restrict__ noise,
                                    float c0, float c1, float c2, float c3, int mode, int clip, int last, long total) {
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT)
    z[i] = sampler_update(z[i], pred[i], noise, i, c0, c1, c2, c3, mode, clip, last);
}

__global__ void cfg_combine_kernel(const float* __restrict__ u, const float* __restrict__ c, float* __restrict__ out,
                                   float s, long total) {
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT)
    out[i] = __fadd_rn(u[i], __fmul_rn(s, __fsub_rn(c[i], u[i])));
}

inline int grid_for(long total) {
  long g = (total + NT - 1) / NT;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int dm_diffuse(const float* x, const float* eps, const int64_t* t, const double* tab64, const float* tab32,
                          float* x_t, int B, long n, int mode, void* stream) {
  DM_REQUIRE(x && eps && t && x_t && B > 0 && n > 0, DM_ERR_ARG, "dm_diffuse: bad argument");
  DM_REQUIRE((mode == 0 && tab64) || (mode == 1 && tab32), DM_ERR_ARG, "dm_diffuse: mode %d needs its schedule table", mode);
  DM_REQUIRE(B <= 65535, DM_ERR_UNSUPPORTED, "dm_diffuse: batch %d > 65535", B);
  int gx = (int)((n + NT - 1) / NT);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(diffuse_kernel, dim3(gx, B), dim3(NT), 0, (hipStream_t)stream, x, eps, t, tab64, tab32, x_t, n, mode);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_mse_partials(void) { return NPART; }

extern "C" int dm_mse_loss_fwd(const float* pred, const float* x, const float* eps, float* partials, float* loss,
                               long total, int mode, void* stream) {
  DM_REQUIRE(pred && eps && partials && loss && total > 0, DM_ERR_ARG, "dm_mse_loss_fwd: bad argument");
  DM_REQUIRE(mode == 0 || (mode == 1 && x), DM_ERR_ARG, "dm_mse_loss_fwd: mode %d needs x", mode);
  int g = grid_for(total);
  if (g > NPART) g = NPART;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(g), dim3(NT), 0, (hipStream_t)stream, pred, x, eps, partials, total, mode);
  DM_CHECK_LAUNCH();
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partials, g, loss, 1.0 / (double)total);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_mse_loss_bwd(const float* pred, const float* x, const float* eps, const float* gout, float* dpred,
                               long total, int mode, void* stream) {
  DM_REQUIRE(pred && eps && gout && dpred && total > 0, DM_ERR_ARG, "dm_mse_loss_bwd: bad argument");
  DM_REQUIRE(mode == 0 || (mode == 1 && x), DM_ERR_ARG, "dm_mse_loss_bwd: mode %d needs x", mode);
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, pred, x, eps, gout, dpred,
                     total, mode, (float)(2.0 / (double)total));
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_sampler_step(float* z, const float* pred, const float* noise, float c0, float c1, float c2, float c3,
                               int mode, int clip, int last, long total, void* stream) {
  DM_REQUIRE(z && pred && total > 0 && mode >= 0 && mode <= 2, DM_ERR_ARG, "dm_sampler_step: bad argument");
  hipLaunchKernelGGL(sampler_step_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, z, pred, noise, c0, c1,
                     c2, c3, mode, clip, last, total);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_sampler_step_dev(float* z, const float* pred, const float* noise, const float* coefs,
                                   long long* step, const int64_t* t_table, int64_t* t_out, int n_steps, int B, int mode,
                                   int clip, long total, void* stream) {
  DM_REQUIRE(z && pred && coefs && step && t_table && t_out && total > 0 && n_steps > 0 && B > 0 && mode >= 0 && mode <= 2,
             DM_ERR_ARG, "dm_sampler_step_dev: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sampler_step_dev_kernel, dim3(grid_for(total)), dim3(NT), 0, st, z, pred, noise, coefs, step, n_steps,
                     mode, clip, total);
  DM_CHECK_LAUNCH();
  hipLaunchKernelGGL(sampler_tick_kernel, dim3(1), dim3(NT), 0, st, step, t_table, t_out, n_steps, B);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_cfg_combine(const float* pred_uncond, const float* pred_cond, float* out, float s, long total,
                              void* stream) {
  DM_REQUIRE(pred_uncond && pred_cond && out && total > 0, DM_ERR_ARG, "dm_cfg_combine: bad argument");
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, pred_uncond, pred_cond,
                     out, s, total);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
