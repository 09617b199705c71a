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
// The products are oriented so that P and dS never leave the registers:
//   forward / dq (a wave owns 64 queries): S^T tile = K_t . Q^T comes out with the lane's column = its query and
//   4 keys per lane - exactly the B operand of O^T += V_t^T . P_t^T (and of dQ^T += K_t^T . dS_t^T) - so the softmax
//   statistics of a query are lane-local up to two shuffles.
//   dk/dv (a wave owns 64 keys): S tile = Q_t . K^T comes out with the lane's column = its key and 4 queries per
//   lane, the B operand of dV^T += dO_t^T . P_t and dK^T += Q_t^T . dS_t.
// A (sample, head) problem is tiny (L = 64, D = 8: 1 KiB each of Q, K, V) and the kernels are bound by how the rows
// are fetched, not by arithmetic: ONE WAVE owns a (sample, head, block of 64 rows); lane r fetches row r of each
// 64-row chunk with 16-byte loads (the four waves of a workgroup take four neighbouring heads, which are contiguous
// in a row of the packed tensor) and stages it in a wave-private piece of LDS, once row-major and, where a product
// needs the transposed operand, once transposed. Fragments are then 8-byte LDS reads. No workgroup barrier: a
// wave's LDS operations execute in order.
typedef s16x4_t frag_t;
constexpr int AT_DP = 16;   // row pitch (elements) of a row-major tile
constexpr int AT_TP = 72;   // row pitch of a transposed tile: 64 rows + 8, fragment reads conflict-free

__device__ __forceinline__ f32x4_t mfma16(frag_t a, frag_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ frag_t pack_frag(float a, float b, float c, float d) {
  const bf16x4_t v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
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
struct AtRow { bf16x8_t v[D / 8]; };

template <int D>
__device__ __forceinline__ AtRow<D> at_load_row(const bf16_t* base, size_t rs, int row, int L) {
  AtRow<D> r;
#pragma unroll
  for (int i = 0; i < D / 8; ++i) r.v[i] = bf16x8_t{};
  if (row < L) {
#pragma unroll
    for (int i = 0; i < D / 8; ++i) r.v[i] = *(const bf16x8_t*)(base + (size_t)row * rs + 8 * i);
  }
  return r;
}
template <int D>
__device__ __forceinline__ void at_store_row(bf16_t* X, int r, const AtRow<D>& row) {
#pragma unroll
  for (int i = 0; i < D / 8; ++i) *(bf16x8_t*)(X + r * AT_DP + 8 * i) = row.v[i];
}
template <int D>
__device__ __forceinline__ void at_store_col(bf16_t* XT, int r, const AtRow<D>& row) {
#pragma unroll
  for (int i = 0; i < D / 8; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) XT[(8 * i + e) * AT_TP + r] = row.v[i][e];
}
// head-dim entries 4g..4g+3 of row `row` of a row-major tile
template <int D>
__device__ __forceinline__ frag_t at_row_frag(const bf16_t* X, int row, int g) {
  if (4 * g < D) return *(const frag_t*)(X + row * AT_DP + 4 * g);
  return frag_t{0, 0, 0, 0};
}
// rows col0 + 4g .. + 3 of head-dim entry d, from a transposed tile
template <int D>
__device__ __forceinline__ frag_t at_col_frag(const bf16_t* XT, int d, int col0, int g) {
  if (d < D) return *(const frag_t*)(XT + d * AT_TP + col0 + 4 * g);
  return frag_t{0, 0, 0, 0};
}
__device__ __forceinline__ void at_lds_fence() {   // compiler-only: a wave's LDS operations execute in order
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// (sample, head, 64-row block) of this wave; false if the wave has nothing to do
__device__ __forceinline__ bool at_wave_problem(int H, int& b, int& h) {
  const int hb = (H + 3) >> 2;
  b = blockIdx.x / hb;
  h = (blockIdx.x - b * hb) * 4 + (threadIdx.x >> 6);
  return h < H;
}

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int L, int H, float scale) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[4][2 * 64 * AT_DP + 16 * AT_TP];
  int b, h;
  if (!at_wave_problem(H, b, h)) return;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  bf16_t* Qs = lds[threadIdx.x >> 6];
  bf16_t* Ks = Qs + 64 * AT_DP;
  bf16_t* Vt = Ks + 64 * AT_DP;
  const int q0 = blockIdx.y * 64;
  const size_t rs = (size_t)3 * H * D;
  const bf16_t* qb = qkv + (size_t)b * L * rs + (size_t)h * D;
  const bf16_t* kb = qb + (size_t)H * D;
  const bf16_t* vb = qb + (size_t)2 * H * D;
  at_store_row<D>(Qs, lane, at_load_row<D>(qb, rs, q0 + lane, L));
  at_lds_fence();
  frag_t qf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) qf[t] = at_row_frag<D>(Qs, 16 * t + j, g);
  f32x4_t o[4];
  float m[4], l[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { o[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; m[t] = -INFINITY; l[t] = 0.f; }
  for (int k0 = 0; k0 < L; k0 += 64) {
    const AtRow<D> kr = at_load_row<D>(kb, rs, k0 + lane, L), vr = at_load_row<D>(vb, rs, k0 + lane, L);
    at_lds_fence();   // the previous chunk's fragment reads are done
    at_store_row<D>(Ks, lane, kr);
    at_store_col<D>(Vt, lane, vr);
    at_lds_fence();
    frag_t kf[4], vf[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      kf[kt] = at_row_frag<D>(Ks, 16 * kt + j, g);
      vf[kt] = at_col_frag<D>(Vt, j, 16 * kt, g);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {     // query tile t: query q0 + 16 t + j in this lane's column
      f32x4_t s[4];
      float cmax = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s[kt] = mfma16(kf[kt], qf[t], f32x4_t{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kt][r] = (k0 + 16 * kt + 4 * g + r < L) ? s[kt][r] * scale : -INFINITY;
          cmax = fmaxf(cmax, s[kt][r]);
        }
      }
      const float mn = fmaxf(m[t], group_max(cmax));   // finite: the chunk holds at least one key
      const float alpha = __expf(m[t] - mn);           // first chunk: exp(-inf) = 0
      float psum = 0.f;
      o[t] *= alpha;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { pr[r] = __expf(s[kt][r] - mn); psum += pr[r]; }
        o[t] = mfma16(vf[kt], pack_frag(pr[0], pr[1], pr[2], pr[3]), o[t]);
      }
      l[t] = l[t] * alpha + group_sum(psum);
      m[t] = mn;
    }
  }
  // o[t]: query q0 + 16 t + j, head-dim entries 4g..4g+3
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int q = q0 + 16 * t + j;
    if (q < L) {
      if (4 * g < D) {
        const float inv = 1.0f / l[t];
        *(frag_t*)(out + ((size_t)b * L + q) * H * D + (size_t)h * D + 4 * g) =
            pack_frag(o[t][0] * inv, o[t][1] * inv, o[t][2] * inv, o[t][3] * inv);
      }
      if (g == 0) lse[((size_t)b * H + h) * L + q] = m[t] + __logf(l[t]);
    }
  }
}

// delta of the row held by this lane: sum_d dout[d] * out[d]
template <int D>
__device__ __forceinline__ float at_row_dot(const AtRow<D>& x, const AtRow<D>& y) {
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < D / 8; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += (float)x.v[i][e] * (float)y.v[i][e];
  return acc;
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                               const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                               bf16_t* __restrict__ dqkv, int L, int H, float scale) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[4][4 * 64 * AT_DP + 16 * AT_TP];
  __shared__ float stat[4][2][64];
  int b, h;
  if (!at_wave_problem(H, b, h)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  bf16_t* Qs = lds[wave];
  bf16_t* Gs = Qs + 64 * AT_DP;
  bf16_t* Ks = Gs + 64 * AT_DP;
  bf16_t* Vs = Ks + 64 * AT_DP;
  bf16_t* Kt = Vs + 64 * AT_DP;
  const int q0 = blockIdx.y * 64;
  const size_t rs = (size_t)3 * H * D, os = (size_t)H * D;
  const bf16_t* qb = qkv + (size_t)b * L * rs + (size_t)h * D;
  const bf16_t* kb = qb + (size_t)H * D;
  const bf16_t* vb = qb + (size_t)2 * H * D;
  const bf16_t* ob = out + (size_t)b * L * os + (size_t)h * D;
  const bf16_t* gb = dout + (size_t)b * L * os + (size_t)h * D;
  {
    const AtRow<D> gr = at_load_row<D>(gb, os, q0 + lane, L);
    at_store_row<D>(Qs, lane, at_load_row<D>(qb, rs, q0 + lane, L));
    at_store_row<D>(Gs, lane, gr);
    stat[wave][0][lane] = at_row_dot<D>(gr, at_load_row<D>(ob, os, q0 + lane, L));
    stat[wave][1][lane] = q0 + lane < L ? lse[((size_t)b * H + h) * L + q0 + lane] : 0.f;
  }
  at_lds_fence();
  frag_t qf[4], gf[4];
  float delta[4], ls[4];
  f32x4_t dq[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    qf[t] = at_row_frag<D>(Qs, 16 * t + j, g);
    gf[t] = at_row_frag<D>(Gs, 16 * t + j, g);
    delta[t] = stat[wave][0][16 * t + j];
    ls[t] = stat[wave][1][16 * t + j];
    dq[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  for (int k0 = 0; k0 < L; k0 += 64) {
    const AtRow<D> kr = at_load_row<D>(kb, rs, k0 + lane, L), vr = at_load_row<D>(vb, rs, k0 + lane, L);
    at_lds_fence();
    at_store_row<D>(Ks, lane, kr);
    at_store_col<D>(Kt, lane, kr);
    at_store_row<D>(Vs, lane, vr);
    at_lds_fence();
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const frag_t kf = at_row_frag<D>(Ks, 16 * kt + j, g), vf = at_row_frag<D>(Vs, 16 * kt + j, g);
      const frag_t ktf = at_col_frag<D>(Kt, j, 16 * kt, g);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t st = mfma16(kf, qf[t], z);    // S^T tile: keys 4g+r, query j
        const f32x4_t dpt = mfma16(vf, gf[t], z);   // dP^T tile
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = (k0 + 16 * kt + 4 * g + r < L) ? __expf(st[r] * scale - ls[t]) : 0.f;
          ds[r] = pr * (dpt[r] - delta[t]) * scale;
        }
        dq[t] = mfma16(ktf, pack_frag(ds[0], ds[1], ds[2], ds[3]), dq[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int q = q0 + 16 * t + j;
    if (q < L && 4 * g < D)
      *(frag_t*)(dqkv + (size_t)b * L * rs + (size_t)q * rs + (size_t)h * D + 4 * g) =
          pack_frag(dq[t][0], dq[t][1], dq[t][2], dq[t][3]);
  }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                bf16_t* __restrict__ dqkv, int L, int H, float scale) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[4][4 * 64 * AT_DP + 2 * 16 * AT_TP];
  __shared__ __attribute__((aligned(16))) float stat[4][2][64];
  int b, h;
  if (!at_wave_problem(H, b, h)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  bf16_t* Ks = lds[wave];
  bf16_t* Vs = Ks + 64 * AT_DP;
  bf16_t* Qs = Vs + 64 * AT_DP;
  bf16_t* Gs = Qs + 64 * AT_DP;
  bf16_t* Qt = Gs + 64 * AT_DP;
  bf16_t* Gt = Qt + 16 * AT_TP;
  const int kt0 = blockIdx.y * 64;
  const size_t rs = (size_t)3 * H * D, os = (size_t)H * D;
  const bf16_t* qb = qkv + (size_t)b * L * rs + (size_t)h * D;
  const bf16_t* kb = qb + (size_t)H * D;
  const bf16_t* vb = qb + (size_t)2 * H * D;
  const bf16_t* ob = out + (size_t)b * L * os + (size_t)h * D;
  const bf16_t* gb = dout + (size_t)b * L * os + (size_t)h * D;
  at_store_row<D>(Ks, lane, at_load_row<D>(kb, rs, kt0 + lane, L));
  at_store_row<D>(Vs, lane, at_load_row<D>(vb, rs, kt0 + lane, L));
  at_lds_fence();
  frag_t kf[4], vf[4];   // B operands: column = key
  f32x4_t dk[4], dv[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    kf[kt] = at_row_frag<D>(Ks, 16 * kt + j, g);
    vf[kt] = at_row_frag<D>(Vs, 16 * kt + j, g);
    dk[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    dv[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  for (int q0 = 0; q0 < L; q0 += 64) {
    const AtRow<D> qr = at_load_row<D>(qb, rs, q0 + lane, L), gr = at_load_row<D>(gb, os, q0 + lane, L);
    const float dl = at_row_dot<D>(gr, at_load_row<D>(ob, os, q0 + lane, L));
    const float lq = q0 + lane < L ? lse[((size_t)b * H + h) * L + q0 + lane] : 0.f;
    at_lds_fence();
    at_store_row<D>(Qs, lane, qr);
    at_store_col<D>(Qt, lane, qr);
    at_store_row<D>(Gs, lane, gr);
    at_store_col<D>(Gt, lane, gr);
    stat[wave][0][lane] = dl;
    stat[wave][1][lane] = lq;
    at_lds_fence();
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const frag_t qf = at_row_frag<D>(Qs, 16 * qt + j, g), gf = at_row_frag<D>(Gs, 16 * qt + j, g);
      const frag_t qtf = at_col_frag<D>(Qt, j, 16 * qt, g), gtf = at_col_frag<D>(Gt, j, 16 * qt, g);
      const f32x4_t dl4 = *(const f32x4_t*)&stat[wave][0][16 * qt + 4 * g];   // queries 4g..4g+3 of the tile
      const f32x4_t ls4 = *(const f32x4_t*)&stat[wave][1][16 * qt + 4 * g];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t st = mfma16(qf, kf[kt], z);    // S tile: queries 4g+r, key j
        const f32x4_t dpt = mfma16(gf, vf[kt], z);   // dP tile
        float pr[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = (q0 + 16 * qt + 4 * g + r < L) ? __expf(st[r] * scale - ls4[r]) : 0.f;
          ds[r] = pr[r] * (dpt[r] - dl4[r]) * scale;
        }
        dv[kt] = mfma16(gtf, pack_frag(pr[0], pr[1], pr[2], pr[3]), dv[kt]);
        dk[kt] = mfma16(qtf, pack_frag(ds[0], ds[1], ds[2], ds[3]), dk[kt]);
      }
    }
  }
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = kt0 + 16 * kt + j;
    if (key < L && 4 * g < D) {
      bf16_t* o = dqkv + (size_t)b * L * rs + (size_t)key * rs + (size_t)h * D + 4 * g;
      *(frag_t*)(o + (size_t)H * D) = pack_frag(dk[kt][0], dk[kt][1], dk[kt][2], dk[kt][3]);
      *(frag_t*)(o + (size_t)2 * H * D) = pack_frag(dv[kt][0], dv[kt][1], dv[kt][2], dv[kt][3]);
    }
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
    const dim3 mgrid(B * dm_cdiv(H, 4), dm_cdiv(L, 64));   // a workgroup = four neighbouring heads, one wave each
    if (D == 8) hipLaunchKernelGGL(attn_fwd_mfma_kernel<8>, mgrid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, scale);
    else hipLaunchKernelGGL(attn_fwd_mfma_kernel<16>, mgrid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, lse, L, H, scale);
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
    const dim3 mgrid(B * dm_cdiv(H, 4), dm_cdiv(L, 64));
    if (D == 8) {
      hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<8>, mgrid, dim3(256), 0, st, q_, o_, g_, lse, (bf16_t*)dqkv, L, H, scale);
      hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<8>, mgrid, dim3(256), 0, st, q_, o_, g_, lse, (bf16_t*)dqkv, L, H, scale);
    } else {
      hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<16>, mgrid, dim3(256), 0, st, q_, o_, g_, lse, (bf16_t*)dqkv, L, H, scale);
      hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<16>, mgrid, dim3(256), 0, st, q_, o_, g_, lse, (bf16_t*)dqkv, L, H, scale);
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
