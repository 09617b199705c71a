// Spatial self-attention core for short sequences (L = H*W of an 8x8 / 16x16 feature map) and small heads
// (D = 8 in every shipped config): thousands of independent (sample, head) problems. Head dims 8 and 16 run on the
// matrix cores (second half of this file); the VALU kernels below cover head dims 32 and 64 (and serve as the
// cross-check of the matrix-core ones in the tests). VALU form: one wave owns 64 queries (forward, dq) or 64 keys (dk, dv) of one (sample, head); the other side is
// streamed through LDS in 64-row chunks and read as broadcasts. Softmax statistics in fp32; the backward
// recomputes probabilities from the saved log-sum-exp, so no L x L matrix is ever stored.
//
// qkv is the packed projection output [B][L][3][H][D] (bf16), exactly the memory the reference views as
// (b, l, 3, h, d); out is [B][L][H][D].
#include "dm_common.h"

extern int g_dm_attn_mfma;

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

// ============================================================================================================
// Matrix-core kernels for head dims 8 and 16 (every shipped config has D = 8): v_mfma_f32_16x16x16_bf16 with the head
// dim padded to the instruction's K = 16. Operand layouts of that instruction (lane l, g = l / 16, j = l % 16):
//   A[i][k]: i = j, k = 4g..4g+3      B[k][n]: n = j, k = 4g..4g+3      D[i][n]: n = j, i = 4g..4g+3
// The products are oriented so that no transposition through LDS is ever needed:
//   forward / dq (a wave owns 16 queries): S^T tile = K_t . Q^T comes out with the lane's column = its query and
//   4 keys per lane - exactly the B operand of O^T += V_t^T . P_t^T (and of dQ^T += K_t^T . dS_t^T) - so the softmax
//   statistics of a query are lane-local up to two shuffles and P never leaves the registers.
//   dk/dv (a wave owns 16 keys): S tile = Q_t . K^T comes out with the lane's column = its key and 4 queries per lane,
//   the B operand of dV^T += dO_t^T . P_t and dK^T += Q_t^T . dS_t.
// Q, K, V rows are read straight from global memory (a (sample, head) problem is 3 KB at L = 64 and lives in L2);
// a "row fragment" is 4 consecutive head-dim entries of one row (8 bytes), a "column fragment" one head-dim entry
// of 4 consecutive rows.
typedef s16x4_t frag_t;

__device__ __forceinline__ f32x4_t mfma16(frag_t a, frag_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ frag_t pack_frag(float a, float b, float c, float d) {
  const bf16x4_t v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
  return __builtin_bit_cast(frag_t, v);
}
template <int D>
__device__ __forceinline__ frag_t row_frag(const bf16_t* base, size_t rs, int row, int L, int g) {
  if (row < L && 4 * g < D) return *(const frag_t*)(base + (size_t)row * rs + 4 * g);
  return frag_t{0, 0, 0, 0};
}
template <int D>
__device__ __forceinline__ frag_t col_frag(const bf16_t* base, size_t rs, int row0, int L, int d, int g) {
  bf16x4_t v = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
  if (d < D) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * g + r;
      if (row < L) v[r] = base[(size_t)row * rs + d];
    }
  }
  return __builtin_bit_cast(frag_t, v);
}
__device__ __forceinline__ float group_max(float v) {   // over the 4 lanes that share a column (l % 16)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int L, int H, float scale) {
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int q0 = (blockIdx.y * 4 + wave) * 16;
  if (q0 >= L) return;   // (wave-uniform; the kernel has no workgroup barrier)
  const size_t rs = (size_t)3 * H * D;
  const bf16_t* qb = qkv + (size_t)b * L * rs + (size_t)h * D;
  const bf16_t* kb = qb + (size_t)H * D;
  const bf16_t* vb = qb + (size_t)2 * H * D;
  const frag_t qf = row_frag<D>(qb, rs, q0 + j, L, g);
  f32x4_t o = {0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < L; k0 += 64) {
    f32x4_t s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      s[t] = mfma16(row_frag<D>(kb, rs, k0 + 16 * t + j, L, g), qf, f32x4_t{0.f, 0.f, 0.f, 0.f});
    float cmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = (k0 + 16 * t + 4 * g + r < L) ? s[t][r] * scale : -INFINITY;
        cmax = fmaxf(cmax, s[t][r]);
      }
    const float mn = fmaxf(m, group_max(cmax));   // finite: the chunk holds at least one key
    const float alpha = __expf(m - mn);           // first chunk: exp(-inf) = 0
    float psum = 0.f;
    frag_t pf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float pr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { pr[r] = __expf(s[t][r] - mn); psum += pr[r]; }
      pf[t] = pack_frag(pr[0], pr[1], pr[2], pr[3]);
    }
    l = l * alpha + group_sum(psum);
    o *= alpha;
    m = mn;
#pragma unroll
    for (int t = 0; t < 4; ++t) o = mfma16(col_frag<D>(vb, rs, k0 + 16 * t, L, j, g), pf[t], o);
  }
  // o: query q0 + j, head-dim entries 4g..4g+3
  const int q = q0 + j;
  if (q < L) {
    if (4 * g < D) {
      const float inv = 1.0f / l;
      *(frag_t*)(out + ((size_t)b * L + q) * H * D + (size_t)h * D + 4 * g) = pack_frag(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
    }
    if (g == 0) lse[((size_t)b * H + h) * L + q] = m + __logf(l);
  }
}

// delta[q] = sum_d dout[q][d] * out[q][d] from the lane's 4 head-dim entries, summed over the 4 lanes of the query
template <int D>
__device__ __forceinline__ float delta_of(const bf16_t* ob, const bf16_t* gb, size_t os, int q, int L, int g) {
  float part = 0.f;
  if (q < L && 4 * g < D) {
    const bf16x4_t ov = *(const bf16x4_t*)(ob + (size_t)q * os + 4 * g), gv = *(const bf16x4_t*)(gb + (size_t)q * os + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) part += (float)ov[r] * (float)gv[r];
  }
  return group_sum(part);
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                               const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                               bf16_t* __restrict__ dqkv, int L, int H, float scale) {
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int q0 = (blockIdx.y * 4 + wave) * 16;
  if (q0 >= L) return;
  const size_t rs = (size_t)3 * H * D, os = (size_t)H * D;
  const bf16_t* qb = qkv + (size_t)b * L * rs + (size_t)h * D;
  const bf16_t* kb = qb + (size_t)H * D;
  const bf16_t* vb = qb + (size_t)2 * H * D;
  const bf16_t* ob = out + (size_t)b * L * os + (size_t)h * D;
  const bf16_t* gb = dout + (size_t)b * L * os + (size_t)h * D;
  const int q = q0 + j;
  const frag_t qf = row_frag<D>(qb, rs, q, L, g), gf = row_frag<D>(gb, os, q, L, g);
  const float delta = delta_of<D>(ob, gb, os, q, L, g);
  const float ls = q < L ? lse[((size_t)b * H + h) * L + q] : 0.f;
  f32x4_t dq = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < L; k0 += 16) {
    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
    const f32x4_t st = mfma16(row_frag<D>(kb, rs, k0 + j, L, g), qf, z);    // S^T tile: keys 4g+r, query j
    const f32x4_t dpt = mfma16(row_frag<D>(vb, rs, k0 + j, L, g), gf, z);   // dP^T tile
    float ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pr = (k0 + 4 * g + r < L) ? __expf(st[r] * scale - ls) : 0.f;
      ds[r] = pr * (dpt[r] - delta) * scale;
    }
    dq = mfma16(col_frag<D>(kb, rs, k0, L, j, g), pack_frag(ds[0], ds[1], ds[2], ds[3]), dq);
  }
  if (q < L && 4 * g < D)
    *(frag_t*)(dqkv + (size_t)b * L * rs + (size_t)q * rs + (size_t)h * D + 4 * g) = pack_frag(dq[0], dq[1], dq[2], dq[3]);
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                bf16_t* __restrict__ dqkv, int L, int H, float scale) {
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int kt0 = (blockIdx.y * 4 + wave) * 16;
  if (kt0 >= L) return;
  const size_t rs = (size_t)3 * H * D, os = (size_t)H * D;
  const bf16_t* qb = qkv + (size_t)b * L * rs + (size_t)h * D;
  const bf16_t* kb = qb + (size_t)H * D;
  const bf16_t* vb = qb + (size_t)2 * H * D;
  const bf16_t* ob = out + (size_t)b * L * os + (size_t)h * D;
  const bf16_t* gb = dout + (size_t)b * L * os + (size_t)h * D;
  const int key = kt0 + j;
  const frag_t kf = row_frag<D>(kb, rs, key, L, g), vf = row_frag<D>(vb, rs, key, L, g);   // B operands: column = key
  f32x4_t dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
  for (int q0 = 0; q0 < L; q0 += 16) {
    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
    const f32x4_t st = mfma16(row_frag<D>(qb, rs, q0 + j, L, g), kf, z);    // S tile: queries 4g+r, key j
    const f32x4_t dpt = mfma16(row_frag<D>(gb, os, q0 + j, L, g), vf, z);   // dP tile
    float pr[4], ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + 4 * g + r;
      float delta = 0.f, ls = 0.f;
      if (q < L) {
        ls = lse[((size_t)b * H + h) * L + q];
#pragma unroll
        for (int d = 0; d < D; d += 8) {
          const bf16x8_t ov = *(const bf16x8_t*)(ob + (size_t)q * os + d), gv = *(const bf16x8_t*)(gb + (size_t)q * os + d);
#pragma unroll
          for (int e = 0; e < 8; ++e) delta += (float)ov[e] * (float)gv[e];
        }
      }
      pr[r] = q < L ? __expf(st[r] * scale - ls) : 0.f;
      ds[r] = pr[r] * (dpt[r] - delta) * scale;
    }
    dv = mfma16(col_frag<D>(gb, os, q0, L, j, g), pack_frag(pr[0], pr[1], pr[2], pr[3]), dv);
    dk = mfma16(col_frag<D>(qb, rs, q0, L, j, g), pack_frag(ds[0], ds[1], ds[2], ds[3]), dk);
  }
  if (key < L && 4 * g < D) {
    bf16_t* o = dqkv + (size_t)b * L * rs + (size_t)key * rs + (size_t)h * D + 4 * g;
    *(frag_t*)(o + (size_t)H * D) = pack_frag(dk[0], dk[1], dk[2], dk[3]);
    *(frag_t*)(o + (size_t)2 * H * D) = pack_frag(dv[0], dv[1], dv[2], dv[3]);
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
  if (g_dm_attn_mfma && (D == 8 || D == 16)) {
    if (D == 8) hipLaunchKernelGGL(attn_fwd_mfma_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, scale);
    else hipLaunchKernelGGL(attn_fwd_mfma_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, scale);
    DM_CHECK_LAUNCH();
    return DM_OK;
  }
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
  if (g_dm_attn_mfma && (D == 8 || D == 16)) {
    hipStream_t st = (hipStream_t)stream;
    const bf16_t *q_ = (const bf16_t*)qkv, *o_ = (const bf16_t*)out, *g_ = (const bf16_t*)dout;
    if (D == 8) {
      hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<8>, grid, dim3(256), 0, st, q_, o_, g_, lse, (bf16_t*)dqkv, L, H, scale);
      hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<8>, grid, dim3(256), 0, st, q_, o_, g_, lse, (bf16_t*)dqkv, L, H, scale);
    } else {
      hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<16>, grid, dim3(256), 0, st, q_, o_, g_, lse, (bf16_t*)dqkv, L, H, scale);
      hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<16>, grid, dim3(256), 0, st, q_, o_, g_, lse, (bf16_t*)dqkv, L, H, scale);
    }
    DM_CHECK_LAUNCH();
    return DM_OK;
  }
  DM_DISPATCH_D(D, hipLaunchKernelGGL(attn_bwd_dq_kernel<DD>, grid, dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                                      (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, L, H, scale));
  DM_CHECK_LAUNCH();
  DM_DISPATCH_D(D, hipLaunchKernelGGL(attn_bwd_dkv_kernel<DD>, grid, dim3(64), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                                      (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, L, H, scale));
  DM_CHECK_LAUNCH();
  return DM_OK;
}
