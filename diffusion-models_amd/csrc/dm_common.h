// Shared device/host helpers for the gfx950 kernels of libdm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dm_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

#define DM_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define DM_OOB 0x80000000u  // byte offset beyond any descriptor range: the load returns zeros

// thread-local last error text; every entry point returns 0 or a dm_status code
void dm_set_error(const char* fmt, ...);

#define DM_REQUIRE(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      dm_set_error(__VA_ARGS__);               \
      return (code);                           \
    }                                          \
  } while (0)

#define DM_CHECK_LAUNCH()                                              \
  do {                                                                 \
    hipError_t e_ = hipGetLastError();                                 \
    if (e_ != hipSuccess) {                                            \
      dm_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
      return DM_ERR_LAUNCH;                                            \
    }                                                                  \
  } while (0)

static inline int dm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Ordered sum of per-workgroup partial vectors into accumulating outputs (dm_sum_item, see the public header): the
// library's replacement for float atomics on parameter gradients. With `deferred` the item is handed to the caller
// (who runs it later through dm_ordered_sum_group) instead of being launched.
typedef dm_sum_item DmOrderedSum;
int dm_finish_ordered_sum(const DmOrderedSum& a, dm_sum_item* deferred, hipStream_t st);

// sigmoid through the hardware reciprocal (1 ulp) instead of an IEEE division (a dozen instructions): these run once
// per element of HBM-bound kernels whose results are rounded to bf16 anyway
__device__ __forceinline__ float dm_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float dm_silu(float v) { return v * dm_sigmoid(v); }
// d/dv [v * sigmoid(v)] = s * (1 + v * (1 - s))
__device__ __forceinline__ float dm_silu_grad(float v) {
  const float s = dm_sigmoid(v);
  return s * (1.0f + v * (1.0f - s));
}

__device__ __forceinline__ float dm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void dm_unpack8(const bf16x8_t& v, float* f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
}
__device__ __forceinline__ bf16x8_t dm_pack8(const float* f) {
  bf16x8_t v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (bf16_t)f[j];
  return v;
}

// XCD-aware tile order: blocks b and b+8 run on the same XCD (private L2), so give each XCD a
// contiguous run of the "minor" index. Returns (major, minor) for a grid of nmajor_padded*nminor blocks,
// nmajor_padded a multiple of 8; callers skip major >= nmajor.
__device__ __forceinline__ void dm_xcd_tile(int bid, int nminor, int& major, int& minor) {
  int xcd = bid & 7, local = bid >> 3;
  major = (local / nminor) * 8 + xcd;
  minor = local % nminor;
}
