// Spatial self-attention core for short sequences (L = H*W of an 8x8 / 16x16 feature map) and small heads
// (D = 8 in every shipped config): thousands of independent (sample, head) problems that fit in registers
// and LDS. One wave owns 64 queries (forward, dq) or 64 keys (dk, dv) of one (sample, head); the other side is
// streamed through LDS in 64-row chunks and read as broadcasts. Softmax statistics in fp32; the backward
// recomputes probabilities from the saved log-sum-exp, so no L x L matrix is ever stored.
//
// qkv is the packed projection output [B][L][3][H][D] (bf16), exactly the memory the reference views as
// (b, l, 3, h, d); out is [B][L][H][D].
#include "dm_common.h"

namespace {

template <int D>
struct RowIO {
  static __device__ __forceinline__ void load(const bf16_t* p, float* f) {
#pragma unroll
    for (int j = 0; j < D; j += 8) dm_unpack8(*(const bf16x8_t*)(p + j), f + j);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* f) {
#pragma unroll
    for (int j = 0; j < D; j += 8) *(bf16x8_t*)(p + j) = dm_pack8(f + j);
  }
};

template <int D>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                      float* __restrict__ lse, int L, int H, float scale) {
  __shared__ __attribute__((aligned(16))) float sk[64][D];
  __shared__ __attribute__((aligned(16))) float sv[64][D];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x;
  const int qi = blockIdx.y * 64 + lane;
  const size_t rs = (size_t)3 * H * D;  // elements per (b, l) row of qkv
  const bf16_t* base = qkv + (size_t)b * L * rs + (size_t)h * D;
  float q[D], o[D];
#pragma unroll
  for (int j = 0; j < D; ++j) { q[j] = 0.f; o[j] = 0.f; }
  if (qi < L) RowIO<D>::load(base + (size_t)qi * rs, q);
#pragma unroll
  for (int j = 0; j < D; ++j) q[j] *= scale;
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < L; k0 += 64) {
    const int kj = k0 + lane;
    float kr[D], vr[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { kr[j] = 0.f; vr[j] = 0.f; }
    if (kj < L) {
      RowIO<D>::load(base + (size_t)kj * rs + (size_t)H * D, kr);
      RowIO<D>::load(base + (size_t)kj * rs + (size_t)2 * H * D, vr);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < D; ++j) { sk[lane][j] = kr[j]; sv[lane][j] = vr[j]; }
    __syncthreads();
    const int nk = min(64, L - k0);
    float mx = m;
    for (int j = 0; j < nk; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) s += q[d] * sk[j][d];
      mx = fmaxf(mx, s);
    }
    const float alpha = __expf(m - mx);  // first chunk: exp(-inf) = 0
    l *= alpha;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] *= alpha;
    for (int j = 0; j < nk; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) s += q[d] * sk[j][d];
      const float p = __expf(s - mx);
      l += p;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] += p * sv[j][d];
    }
    m = mx;
  }
  if (qi < L) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] *= inv;
    RowIO<D>::store(out + ((size_t)b * L + qi) * H * D + (size_t)h * D, o);
    lse[((size_t)b * H + h) * L + qi] = m + __logf(l);
  }
}

// dq: one lane per query
template <int D>
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                         const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                         bf16_t* __restrict__ dqkv, int L, int H, float scale) {
  __shared__ __attribute__((aligned(16))) float sk[64][D];
  __shared__ __attribute__((aligned(16))) float sv[64][D];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x;
  const int qi = blockIdx.y * 64 + lane;
  const size_t rs = (size_t)3 * H * D;
  const bf16_t* base = qkv + (size_t)b * L * rs + (size_t)h * D;
  float q[D], go[D], dq[D];
  float delta = 0.f, ls = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) { q[j] = 0.f; go[j] = 0.f; dq[j] = 0.f; }
  if (qi < L) {
    float oo[D];
    RowIO<D>::load(base + (size_t)qi * rs, q);
    const size_t orow = ((size_t)b * L + qi) * H * D + (size_t)h * D;
    RowIO<D>::load(dout + orow, go);
    RowIO<D>::load(out + orow, oo);
#pragma unroll
    for (int j = 0; j < D; ++j) delta += go[j] * oo[j];
    ls = lse[((size_t)b * H + h) * L + qi];
  }
  for (int k0 = 0; k0 < L; k0 += 64) {
    const int kj = k0 + lane;
    float kr[D], vr[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { kr[j] = 0.f; vr[j] = 0.f; }
    if (kj < L) {
      RowIO<D>::load(base + (size_t)kj * rs + (size_t)H * D, kr);
      RowIO<D>::load(base + (size_t)kj * rs + (size_t)2 * H * D, vr);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < D; ++j) { sk[lane][j] = kr[j]; sv[lane][j] = vr[j]; }
    __syncthreads();
    const int nk = min(64, L - k0);
    for (int j = 0; j < nk; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += q[d] * sk[j][d]; dp += go[d] * sv[j][d]; }
      const float p = __expf(s * scale - ls);
      const float ds = p * (dp - delta) * scale;
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] += ds * sk[j][d];
    }
  }
  if (qi < L) RowIO<D>::store(dqkv + (size_t)b * L * rs + (size_t)qi * rs + (size_t)h * D, dq);
}

// dk, dv: one lane per key
template <int D>
__global__ __launch_bounds__(64) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                          const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dqkv, int L, int H, float scale) {
  __shared__ __attribute__((aligned(16))) float sq[64][D];
  __shared__ __attribute__((aligned(16))) float sg[64][D];
  __shared__ float sl[64];
  __shared__ float sd[64];
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x;
  const int kj = blockIdx.y * 64 + lane;
  const size_t rs = (size_t)3 * H * D;
  const bf16_t* base = qkv + (size_t)b * L * rs + (size_t)h * D;
  float k[D], v[D], dk[D], dv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) { k[j] = 0.f; v[j] = 0.f; dk[j] = 0.f; dv[j] = 0.f; }
  if (kj < L) {
    RowIO<D>::load(base + (size_t)kj * rs + (size_t)H * D, k);
    RowIO<D>::load(base + (size_t)kj * rs + (size_t)2 * H * D, v);
  }
  for (int q0 = 0; q0 < L; q0 += 64) {
    const int qi = q0 + lane;
    float qr[D], gr[D], orr[D];
    float delta = 0.f, ls = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) { qr[j] = 0.f; gr[j] = 0.f; }
    if (qi < L) {
      RowIO<D>::load(base + (size_t)qi * rs, qr);
      const size_t orow = ((size_t)b * L + qi) * H * D + (size_t)h * D;
      RowIO<D>::load(dout + orow, gr);
      RowIO<D>::load(out + orow, orr);
#pragma unroll
      for (int j = 0; j < D; ++j) delta += gr[j] * orr[j];
      ls = lse[((size_t)b * H + h) * L + qi];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < D; ++j) { sq[lane][j] = qr[j]; sg[lane][j] = gr[j]; }
    sl[lane] = ls;
    sd[lane] = delta;
    __syncthreads();
    const int nq = min(64, L - q0);
    for (int i = 0; i < nq; ++i) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += sq[i][d] * k[d]; dp += sg[i][d] * v[d]; }
      const float p = __expf(s * scale - sl[i]);
      const float ds = p * (dp - sd[i]) * scale;
#pragma unroll
      for (int d = 0; d < D; ++d) { dv[d] += p * sg[i][d]; dk[d] += ds * sq[i][d]; }
    }
  }
  if (kj < L) {
    bf16_t* o = dqkv + (size_t)b * L * rs + (size_t)kj * rs + (size_t)h * D;
    RowIO<D>::store(o + (size_t)H * D, dk);
    RowIO<D>::store(o + (size_t)2 * H * D, dv);
  }
}

int check(const char* who, const void* a, const void* b, int B, int L, int H, int D) {
  DM_REQUIRE(a && b, DM_ERR_ARG, "%s: null argument", who);
  DM_REQUIRE(B > 0 && L > 0 && H > 0, DM_ERR_SHAPE, "%s: empty problem", who);
  DM_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64, DM_ERR_UNSUPPORTED, "%s: head dim %d not in {8,16,32,64}", who, D);
  DM_REQUIRE((long)B * H < 0x7fffffffL, DM_ERR_UNSUPPORTED, "%s: too many heads", who);
  return DM_OK;
}

}  // namespace

#define DM_DISPATCH_D(D, ...)              \
  switch (D) {                             \
    case 8: { constexpr int DD = 8; __VA_ARGS__; } break;   \
    case 16: { constexpr int DD = 16; __VA_ARGS__; } break; \
    case 32: { constexpr int DD = 32; __VA_ARGS__; } break; \
    default: { constexpr int DD = 64; __VA_ARGS__; } break; \
  }

extern "C" int dm_attention_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int D, float scale,
                                void* stream) {
  int rc = check("dm_attention_fwd", qkv, out, B, L, H, D);
  if (rc) return rc;
  DM_REQUIRE(lse, DM_ERR_ARG, "dm_attention_fwd: null lse");
  dim3 grid(B * H, dm_cdiv(L, 64));
  DM_DISPATCH_D(D, hipLaunchKernelGGL(attn_fwd_kernel<DD>, grid, dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                                      (bf16_t*)out, lse, L, H, scale));
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B,
                                int L, int H, int D, float scale, void* stream) {
  int rc = check("dm_attention_bwd", qkv, out, B, L, H, D);
  if (rc) return rc;
  DM_REQUIRE(dout && lse && dqkv, DM_ERR_ARG, "dm_attention_bwd: null argument");
  dim3 grid(B * H, dm_cdiv(L, 64));
  DM_DISPATCH_D(D, hipLaunchKernelGGL(attn_bwd_dq_kernel<DD>, grid, dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                                      (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, L, H, scale));
  DM_CHECK_LAUNCH();
  DM_DISPATCH_D(D, hipLaunchKernelGGL(attn_bwd_dkv_kernel<DD>, grid, dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                                      (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, L, H, scale));
  DM_CHECK_LAUNCH();
  return DM_OK;
}
