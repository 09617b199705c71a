// Small fp32 operators of the embedding path, resampling helpers, weight packing and column sums.
// All are latency- or HBM-bound and tiny next to the convolutions.
#include "dm_common.h"

namespace {

constexpr int NT = 256;

// ---- sinusoidal timestep encoding: out[b] = [sin(t*freq) | cos(t*freq)]
__global__ void timestep_embed_kernel(const int64_t* __restrict__ t, const float* __restrict__ freq,
                                      float* __restrict__ out, int B, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i - b * half;
  const float a = (float)t[b] * freq[k];
  out[(size_t)b * 2 * half + k] = sinf(a);
  out[(size_t)b * 2 * half + half + k] = cosf(a);
}

// Staging of an fp32 tile into LDS for the embedding-path GEMMs. These launches are small (a few dozen workgroups) and
// sit on the step's critical path at both ends of the network, so what matters is the length of a workgroup's chain of
// memory round trips (each a few microseconds behind the cold TLBs and caches of a step): a thread fetches its share of
// the tile as 16-byte pieces, eight of them in flight at a time, instead of one float per loop iteration (32-48
// dependent trips for a 64 x 128 tile). Tile element (r, c4 * 4 ..) comes from src[r * ld + c4 * 4 ..] when
// r < rows_valid (else zeros); `put(r, c, v)` files the four values. cols must be a multiple of 4 and src + r * ld
// 16-byte aligned (the callers check).
template <typename Put>
__device__ __forceinline__ void stage_tile4(const float* __restrict__ src, long ld, int rows, int cols, int rows_valid,
                                            Put put) {
  const int c4n = cols >> 2, total = rows * c4n;
  for (int base = threadIdx.x; base < total; base += NT * 8) {
    f32x4_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = base + j * NT;
      const int r = i / c4n, c4 = i - r * c4n;
      v[j] = (i < total && r < rows_valid) ? *(const f32x4_t*)(src + (long)r * ld + c4 * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = base + j * NT;
      if (i < total) {
        const int r = i / c4n, c4 = i - r * c4n;
        put(r, c4 * 4, v[j]);
      }
    }
  }
}
__device__ __forceinline__ bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// acc += a * b as ONE v_fmac_f32, spelled out. Left to the compiler, the forward kernels' inner loops become packed
// v_pk_fma_f32 on pairs of rows, and for the odd elements of a 16-byte LDS read in the form `op_sel:[0,1,0]` (the low
// lane multiplies by the HIGH register of the second source pair). On MI355X that form - and v_pk_mul_f32 /
// v_pk_add_f32 with op_sel:[0,1]; no other operand-select form - loses the low lane's product in lanes 48-63 now and
// then while waves of ANOTHER kernel that issue matrix-core instructions and direct-to-LDS buffer loads share the
// compute unit (this library's 3x3 convolutions and the vendor's 16-bit GEMMs are such kernels): one term
// x[m][k] * w[n][k] missing from 16 neighbouring outputs of an even row, in 80 % of the launches beside a bf16 GEMM and
// in a few percent of training steps; the same loop with plain FMAs never did (tools/experiments/pk_forms_repro.hip is
// a stand-alone reproducer, repro_linear_group.py the one with these kernels). The loops are bound by their LDS
// reads, so the unpacked form costs nothing. It also fixes the contraction: a row's result must not depend on its
// place in the tile.
__device__ __forceinline__ void fma_acc(float& acc, float a, float b) {
  asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}

// ---- y[m][n] = sum_k f(x[m][k]) w[n][k] + bias[n]; block = 64 columns x 16 rows, K walked through LDS in chunks of at
// most LIN_KC (one chunk for the widths the UNet configurations use; the sum runs over k in ascending order either way)
constexpr int LIN_ROWS = 16, LIN_COLS = 64, LIN_KC = 128;
__device__ __forceinline__ void linear_tile(const float* __restrict__ x, const float* __restrict__ w,
                                            const float* __restrict__ bias, float* __restrict__ y, int M, int K, int N,
                                            int n0, int m0, int act_in) {
  extern __shared__ float lds[];
  const int kc = min(K, LIN_KC);
  float* xs = lds;                     // [LIN_ROWS][kc]
  float* ws = lds + LIN_ROWS * kc;     // [LIN_COLS][kc+4]: 16-byte reads along k, eight lanes cover the 32 banks
  const bool vec = K % 4 == 0 && aligned16(x) && aligned16(w);
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;  // 4 row groups of 4 rows
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += kc) {
    const int kn = min(kc, K - k0);
    if (k0) __syncthreads();
    if (vec) {
      stage_tile4(x + (size_t)m0 * K + k0, K, LIN_ROWS, kn, M - m0, [&](int r, int k, f32x4_t v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[r * kc + k + e] = act_in ? dm_silu(v[e]) : v[e];
      });
      stage_tile4(w + (size_t)n0 * K + k0, K, LIN_COLS, kn, N - n0, [&](int cc, int k, f32x4_t v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ws[cc * (kc + 4) + k + e] = v[e];
      });
    } else {
      for (int i = threadIdx.x; i < LIN_ROWS * kn; i += NT) {
        const int r = i / kn, k = i - r * kn;
        float v = (m0 + r < M) ? x[(size_t)(m0 + r) * K + k0 + k] : 0.f;
        xs[r * kc + k] = act_in ? dm_silu(v) : v;
      }
      for (int i = threadIdx.x; i < LIN_COLS * kn; i += NT) {
        const int cc = i / kn, k = i - cc * kn;
        ws[cc * (kc + 4) + k] = (n0 + cc < N) ? w[(size_t)(n0 + cc) * K + k0 + k] : 0.f;
      }
    }
    __syncthreads();
    // (the loop is bound by its LDS reads: four k per 16-byte read. The same fused multiply-adds in the same order in
    // both forms: see fma_acc. kn % 4 == 0 implies kc % 4 == 0, so every row starts on a 16-byte boundary)
    if (kn % 4 == 0) {
      for (int k = 0; k < kn; k += 4) {
        const f32x4_t wv = *(const f32x4_t*)(ws + c * (kc + 4) + k);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4_t xv = *(const f32x4_t*)(xs + (rg * 4 + r) * kc + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) fma_acc(acc[r], xv[e], wv[e]);
        }
      }
    } else {
      for (int k = 0; k < kn; ++k) {
        const float wv = ws[c * (kc + 4) + k];
#pragma unroll
        for (int r = 0; r < 4; ++r) fma_acc(acc[r], xs[(rg * 4 + r) * kc + k], wv);
      }
    }
  }
  if (n0 + c < N) {
    const float bv = bias ? bias[n0 + c] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + rg * 4 + r;
      if (m < M) y[(size_t)m * N + n0 + c] = acc[r] + bv;
    }
  }
}
__host__ __device__ constexpr int linear_tile_lds_bytes(int K) {
  return (LIN_ROWS * (K < LIN_KC ? K : LIN_KC) + LIN_COLS * ((K < LIN_KC ? K : LIN_KC) + 4)) * 4;
}

__global__ __launch_bounds__(NT) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int M,
                                                        int K, int N, int act_in) {
  linear_tile(x, w, bias, y, M, K, N, blockIdx.x * LIN_COLS, blockIdx.y * LIN_ROWS, act_in);
}

// dx[m][k] = f'(x[m][k]) * sum_n dy[m][n] w[n][k]; block = 4 rows x 64 columns, n streamed through LDS in chunks
constexpr int BX_ROWS = 4, BX_COLS = 64, BX_CHUNK = 32;
__global__ __launch_bounds__(NT) void linear_bwd_x_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ dy, float* __restrict__ dx, int M,
                                                          int K, int N, int act_in) {
  __shared__ float sdy[BX_ROWS][BX_CHUNK];
  __shared__ __attribute__((aligned(16))) float sw[BX_CHUNK][BX_COLS];
  const int k0 = blockIdx.x * BX_COLS, m0 = blockIdx.y * BX_ROWS;
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  float acc = 0.f;
  for (int n0 = 0; n0 < N; n0 += BX_CHUNK) {
    __syncthreads();
    if (threadIdx.x < BX_ROWS * BX_CHUNK) {
      const int rr = threadIdx.x / BX_CHUNK, nn = threadIdx.x % BX_CHUNK;
      sdy[rr][nn] = (m0 + rr < M && n0 + nn < N) ? dy[(size_t)(m0 + rr) * N + n0 + nn] : 0.f;
    }
    if (K % 4 == 0 && k0 + BX_COLS <= K && aligned16(w)) {
      stage_tile4(w + (size_t)n0 * K + k0, K, BX_CHUNK, BX_COLS, N - n0, [&](int nn, int cc, f32x4_t v) {
        *(f32x4_t*)&sw[nn][cc] = v;
      });
    } else {
      for (int i = threadIdx.x; i < BX_CHUNK * BX_COLS; i += NT) {
        const int nn = i / BX_COLS, cc = i % BX_COLS;
        sw[nn][cc] = (n0 + nn < N && k0 + cc < K) ? w[(size_t)(n0 + nn) * K + k0 + cc] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int nn = 0; nn < BX_CHUNK; ++nn) acc += sdy[r][nn] * sw[nn][c];
  }
  const int m = m0 + r, k = k0 + c;
  if (m < M && k < K) {
    if (act_in) acc *= dm_silu_grad(x[(size_t)m * K + k]);
    dx[(size_t)m * K + k] = acc;
  }
}

// dw[n][k] += sum_m dy[m][n] f(x[m][k]); dbias[n] += sum_m dy[m][n]; block = 4 output rows x 64 columns,
// the 4 dy columns staged in LDS, x read coalesced along k
constexpr int BW_ROWS = 4, BW_COLS = 64, BW_MAXM = 512;
__global__ __launch_bounds__(NT) void linear_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dw, float* __restrict__ dbias, int M, int K,
                                                          int N, int act_in) {
  __shared__ float sdy[BW_MAXM][BW_ROWS];
  const int k0 = blockIdx.x * BW_COLS, n0 = blockIdx.y * BW_ROWS;
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int k = k0 + c, n = n0 + r;
  float acc = 0.f, bsum = 0.f;
  for (int mb = 0; mb < M; mb += BW_MAXM) {
    const int mc = min(BW_MAXM, M - mb);
    __syncthreads();
    for (int i = threadIdx.x; i < mc * BW_ROWS; i += NT) {
      const int mm = i / BW_ROWS, rr = i % BW_ROWS;
      sdy[mm][rr] = (n0 + rr < N) ? dy[(size_t)(mb + mm) * N + n0 + rr] : 0.f;
    }
    __syncthreads();
    if (k < K) {
      int mm = 0;
      for (; mm + 16 <= mc; mm += 16) {
        float xv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) xv[j] = x[(size_t)(mb + mm + j) * K + k];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += sdy[mm + j][r] * (act_in ? dm_silu(xv[j]) : xv[j]);
      }
      for (; mm < mc; ++mm) {
        float xv = x[(size_t)(mb + mm) * K + k];
        if (act_in) xv = dm_silu(xv);
        acc += sdy[mm][r] * xv;
      }
    }
    if (dbias && blockIdx.x == 0 && c == 0)
      for (int mm = 0; mm < mc; ++mm) bsum += sdy[mm][r];
  }
  if (n < N) {
    if (dw && k < K) dw[(size_t)n * K + k] += acc;
    if (dbias && blockIdx.x == 0 && c == 0) dbias[n] += bsum;
  }
}

// ---- grouped linear: P independent layers y_p = f(x) w_p^T + b_p sharing one input x [M][K] (the 17 emb_proj
// layers of a UNet forward all read SiLU(embedding)); one launch for all of them, forward and backward.
constexpr int MAX_GROUP = 32;
struct LinearGroupItem {   // one per layer; the whole table travels by value in the kernel arguments
  const float* w;   // [N][K]
  const float* b;   // [N] or null
  float* y;         // [M][N]
  const float* dy;  // [M][N]
  float* dw;        // [N][K] accumulates, or null
  float* db;        // [N] accumulates, or null
  int N;
  int tile0;        // first 64-column tile / first 4-row tile of this layer in the launch grid
};

struct LinearGroup {
  LinearGroupItem it[MAX_GROUP];
  int P;
};

__device__ __forceinline__ int find_item(const LinearGroup& g, int tile) {
  int p = 0;
  while (p + 1 < g.P && g.it[p + 1].tile0 <= tile) ++p;
  return p;
}

__global__ __launch_bounds__(NT) void linear_group_fwd_kernel(const float* __restrict__ x, const LinearGroup g, int M, int K,
                                                              int act_in) {
  const LinearGroupItem it = g.it[find_item(g, blockIdx.x)];
  linear_tile(x, it.w, it.b, it.y, M, K, it.N, (blockIdx.x - it.tile0) * LIN_COLS, blockIdx.y * LIN_ROWS, act_in);
}

// Backward of the grouped linear as two register-tiled fp32 GEMMs (the layers differ only in N, so the launch grid
// walks 64-wide slices of all layers' N axes: item.tile0 = first slice of a layer).
constexpr int GT_M = 128, GT_K = 128, GT_N = 64;   // tile of rows (batch), input features, slice of output features
constexpr int GT_PITCH = GT_M + 4;
constexpr int GT_XN = 64;    // output features one workgroup of the input-gradient kernel covers (the layers are few and narrow: more workgroups matter more than fewer partial tiles)

// Input gradient in two steps without atomics (several dozen workgroups adding into one 256 KB dx through float
// atomics ran at a tenth of the atomic rate: every adder hits the same rows):
//   part[s][m][k] = sum_{n in slice group s} dy_p[m][n] w_p[n][k]      (plain stores into the workspace)
//   dx[m][k]      = f'(x[m][k]) * sum_s part[s][m][k]                  (fixed order)
__global__ __launch_bounds__(NT) void linear_group_bwd_x_kernel(const LinearGroup g, float* __restrict__ part, int M, int K) {
  __shared__ __attribute__((aligned(16))) float sdy[GT_N * GT_PITCH];  // [n][m], transposed
  __shared__ __attribute__((aligned(16))) float sw[GT_N * GT_K];       // [n][k]
  const LinearGroupItem it = g.it[find_item(g, blockIdx.x)];
  const int N = it.N;
  const int nb = (blockIdx.x - it.tile0) * GT_XN, m0 = blockIdx.y * GT_M, k0 = blockIdx.z * GT_K;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;   // rows ty*8.., columns tx*8..
  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  for (int n0 = nb; n0 < min(N, nb + GT_XN); n0 += GT_N) {
    __syncthreads();
    if (N % 4 == 0 && K % 4 == 0 && n0 + GT_N <= N && k0 + GT_K <= K && aligned16(it.dy) && aligned16(it.w)) {
      stage_tile4(it.dy + (size_t)m0 * N + n0, N, GT_M, GT_N, M - m0, [&](int mm, int nn, f32x4_t v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sdy[(nn + e) * GT_PITCH + mm] = v[e];
      });
      stage_tile4(it.w + (size_t)n0 * K + k0, K, GT_N, GT_K, GT_N, [&](int nn, int kk, f32x4_t v) {
        *(f32x4_t*)(sw + nn * GT_K + kk) = v;
      });
    } else {
      for (int i = threadIdx.x; i < GT_M * GT_N; i += NT) {
        const int mm = i / GT_N, nn = i - mm * GT_N;
        sdy[nn * GT_PITCH + mm] = (m0 + mm < M && n0 + nn < N) ? it.dy[(size_t)(m0 + mm) * N + n0 + nn] : 0.f;
      }
      for (int i = threadIdx.x; i < GT_N * GT_K; i += NT) {
        const int nn = i / GT_K, kk = i - nn * GT_K;
        sw[i] = (n0 + nn < N && k0 + kk < K) ? it.w[(size_t)(n0 + nn) * K + k0 + kk] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int nn = 0; nn < GT_N; ++nn) {
      const f32x4_t d0 = *(const f32x4_t*)(sdy + nn * GT_PITCH + ty * 8), d1 = *(const f32x4_t*)(sdy + nn * GT_PITCH + ty * 8 + 4);
      const f32x4_t w0 = *(const f32x4_t*)(sw + nn * GT_K + tx * 8), w1 = *(const f32x4_t*)(sw + nn * GT_K + tx * 8 + 4);
      const float dv[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
      const float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] += dv[i] * wv[j];
    }
  }
  // the 16 threads of a row group cover 512 contiguous bytes of each row
  float* out = part + (size_t)blockIdx.x * M * K;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + ty * 8 + i, k = k0 + tx * 8;
    if (m < M) {
      if (k + 8 <= K && K % 4 == 0) {
        *(f32x4_t*)(out + (size_t)m * K + k) = f32x4_t{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        *(f32x4_t*)(out + (size_t)m * K + k + 4) = f32x4_t{acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k + j < K) out[(size_t)m * K + k + j] = acc[i][j];
      }
    }
  }
}

// dx = f'(x) * sum over the slice groups, in group order: a workgroup owns 64 elements, wave g sums groups g, g+4, ...
__global__ __launch_bounds__(NT) void linear_group_bwd_x_sum_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                                    float* __restrict__ dx, int total, int nparts, int act_in) {
  __shared__ float red[3][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (e < total) {
    int q = grp;
    for (; q + 28 < nparts; q += 32) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(q + 4 * j) * total + e];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; q < nparts; q += 4) s += part[(size_t)q * total + e];
  }
  if (grp) red[grp - 1][lane] = s;
  __syncthreads();
  if (grp || e >= total) return;
  s = ((s + red[0][lane]) + red[1][lane]) + red[2][lane];
  if (act_in) s *= dm_silu_grad(x[e]);
  dx[e] = s;
}

// dw_p[n][k] += sum_m dy_p[m][n] f(x[m][k]); db_p[n] += sum_m dy_p[m][n]; one workgroup owns a 64 x 128 tile of dw_p
__global__ __launch_bounds__(NT) void linear_group_bwd_w_kernel(const float* __restrict__ x, const LinearGroup g, int M, int K,
                                                                int act_in) {
  __shared__ __attribute__((aligned(16))) float sdy[GT_N * GT_N];   // [m (64)][n (64)]
  __shared__ __attribute__((aligned(16))) float sx[GT_N * GT_K];    // [m (64)][k (128)]
  const LinearGroupItem it = g.it[find_item(g, blockIdx.x)];
  const int N = it.N;
  const int n0 = (blockIdx.x - it.tile0) * GT_N, k0 = blockIdx.y * GT_K;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;   // rows (n) ty*4.., columns (k) tx*8..
  float acc[4][8], bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  for (int mb = 0; mb < M; mb += GT_N) {
    __syncthreads();
    if (N % 4 == 0 && K % 4 == 0 && n0 + GT_N <= N && k0 + GT_K <= K && aligned16(it.dy) && aligned16(x)) {
      stage_tile4(it.dy + (size_t)mb * N + n0, N, GT_N, GT_N, M - mb, [&](int mm, int nn, f32x4_t v) {
        *(f32x4_t*)(sdy + mm * GT_N + nn) = v;
      });
      stage_tile4(x + (size_t)mb * K + k0, K, GT_N, GT_K, M - mb, [&](int mm, int kk, f32x4_t v) {
        if (act_in) v = f32x4_t{dm_silu(v[0]), dm_silu(v[1]), dm_silu(v[2]), dm_silu(v[3])};
        *(f32x4_t*)(sx + mm * GT_K + kk) = v;
      });
    } else {
      for (int i = threadIdx.x; i < GT_N * GT_N; i += NT) {
        const int mm = i / GT_N, nn = i - mm * GT_N;
        sdy[i] = (mb + mm < M && n0 + nn < N) ? it.dy[(size_t)(mb + mm) * N + n0 + nn] : 0.f;
      }
      for (int i = threadIdx.x; i < GT_N * GT_K; i += NT) {
        const int mm = i / GT_K, kk = i - mm * GT_K;
        float v = (mb + mm < M && k0 + kk < K) ? x[(size_t)(mb + mm) * K + k0 + kk] : 0.f;
        sx[i] = act_in ? dm_silu(v) : v;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int mm = 0; mm < GT_N; ++mm) {
      const f32x4_t d = *(const f32x4_t*)(sdy + mm * GT_N + ty * 4);
      const f32x4_t x0 = *(const f32x4_t*)(sx + mm * GT_K + tx * 8), x1 = *(const f32x4_t*)(sx + mm * GT_K + tx * 8 + 4);
      const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bsum[i] += d[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] += d[i] * xv[j];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty * 4 + i;
    if (n >= N) continue;
    if (it.dw) {
      float* dst = it.dw + (size_t)n * K + k0 + tx * 8;
      if (K % 4 == 0 && k0 + tx * 8 + 8 <= K && aligned16(it.dw)) {
        const f32x4_t o0 = *(const f32x4_t*)dst, o1 = *(const f32x4_t*)(dst + 4);
        *(f32x4_t*)dst = f32x4_t{o0[0] + acc[i][0], o0[1] + acc[i][1], o0[2] + acc[i][2], o0[3] + acc[i][3]};
        *(f32x4_t*)(dst + 4) = f32x4_t{o1[0] + acc[i][4], o1[1] + acc[i][5], o1[2] + acc[i][6], o1[3] + acc[i][7]};
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k0 + tx * 8 + j < K) dst[j] += acc[i][j];
      }
    }
    if (it.db && blockIdx.y == 0 && tx == 0) it.db[n] += bsum[i];
  }
}

// ---- class embedding
__device__ __forceinline__ int class_index(const int64_t* cls, const float* mask, int uncond, int b) {
  if (!cls) return uncond;
  if (mask && mask[b] != 0.f) return uncond;
  return (int)cls[b];
}
__global__ void class_embed_add_kernel(float* __restrict__ emb, const float* __restrict__ table,
                                       const int64_t* __restrict__ cls, const float* __restrict__ mask, int uncond, int B,
                                       int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * E) return;
  const int b = i / E, e = i - b * E;
  emb[i] += table[(size_t)class_index(cls, mask, uncond, b) * E + e];
}
// one thread per (table row, feature): the samples that map to the row are added in batch order
constexpr int CE_CHUNK = 1024;
__global__ void class_embed_bwd_kernel(const float* __restrict__ demb, float* __restrict__ dtable,
                                       const int64_t* __restrict__ cls, const float* __restrict__ mask, int uncond, int B,
                                       int E) {
  __shared__ unsigned char mine[CE_CHUNK];
  const int e = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
  if (row == uncond) return;
  float s = 0.f;
  for (int b0 = 0; b0 < B; b0 += CE_CHUNK) {   // which samples map to this row: looked up once, by all threads together
    const int nb = min(CE_CHUNK, B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) mine[i] = class_index(cls, mask, uncond, b0 + i) == row;
    __syncthreads();
    if (e < E) {
      int i = 0;
      for (; i + 8 <= nb; i += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = mine[i + j] ? demb[(size_t)(b0 + i + j) * E + e] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (mine[i + j]) s += v[j];
      }
      for (; i < nb; ++i)
        if (mine[i]) s += demb[(size_t)(b0 + i) * E + e];
    }
  }
  if (e < E) dtable[(size_t)row * E + e] += s;
}

// ---- 2x resampling, NHWC bf16, one thread per 8 channels of one OUTPUT pixel
__global__ void upsample2x_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int VP) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * 4 * H * W * VP;
  if (i >= total) return;
  const int v = i % VP;
  const long p = i / VP;
  const int ow = p % (2 * W), oh = (p / (2 * W)) % (2 * H), b = p / (4L * H * W);
  *(bf16x8_t*)(y + p * VP * 8 + v * 8) = *(const bf16x8_t*)(x + (((long)b * H + (oh >> 1)) * W + (ow >> 1)) * VP * 8 + v * 8);
}
// dx[h][w] = sum of the 2x2 block of dy; with avg=1 scaled by 1/4 (average pooling forward uses the same gather)
__global__ void gather2x2_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int B, int H, int W, int VP,
                                 float k) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * H * W * VP;
  if (i >= total) return;
  const int v = i % VP;
  const long p = i / VP;
  const int w = p % W, h = (p / W) % H, b = p / ((long)H * W);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      float f[8];
      dm_unpack8(*(const bf16x8_t*)(src + (((long)b * 2 * H + 2 * h + dh) * 2 * W + 2 * w + dw) * VP * 8 + v * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] *= k;
  *(bf16x8_t*)(dst + p * VP * 8 + v * 8) = dm_pack8(acc);
}
// dst (2H x 2W) = k * src (H x W) broadcast over 2x2 blocks (average-pool backward, k = 1/4)
__global__ void scatter2x2_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int B, int H, int W, int VP,
                                  float k) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * 4 * H * W * VP;
  if (i >= total) return;
  const int v = i % VP;
  const long p = i / VP;
  const int ow = p % (2 * W), oh = (p / (2 * W)) % (2 * H), b = p / (4L * H * W);
  float f[8];
  dm_unpack8(*(const bf16x8_t*)(src + (((long)b * H + (oh >> 1)) * W + (ow >> 1)) * VP * 8 + v * 8), f);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] *= k;
  *(bf16x8_t*)(dst + p * VP * 8 + v * 8) = dm_pack8(f);
}

// ---- weight packing: fp32 [Cout][Cin][KH][KW] (strided) -> bf16 [Cout][T][Cin] and bf16 [Cin][T][Cout]
__global__ void pack_weight_kernel(const float* __restrict__ w, long s_co, long s_ci, long s_kh, long s_kw, int Cout,
                                   int Cin, int KH, int KW, bf16_t* __restrict__ wf, bf16_t* __restrict__ wd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int T = KH * KW;
  if (i >= (long)Cout * T * Cin) return;
  const int ci = i % Cin;
  const int tap = (i / Cin) % T;
  const int co = i / ((long)Cin * T);
  const int kh = tap / KW, kw = tap - kh * KW;
  const bf16_t v = (bf16_t)w[co * s_co + ci * s_ci + kh * s_kh + kw * s_kw];
  if (wf) wf[i] = v;
  if (wd) wd[((long)ci * T + tap) * Cout + co] = v;
}

// ---- column sums: per-(row chunk, 64-column slab) partial sums with plain stores, then the ordered sum
// One workgroup = 64 columns (8 vectors of 8) x 32 row lanes over rows [blockIdx.y * rpb, +rpb).
__global__ __launch_bounds__(NT) void colsum_part_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, long M,
                                                         int C, int rows_per_block) {
  __shared__ float red[32][65];
  const int v = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64, c = c0 + 8 * v;
  const long m0 = (long)blockIdx.y * rows_per_block;
  const long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < C) {
    long m = m0 + pl;
    for (; m + 96 < m1; m += 128) {
      const bf16x8_t r0 = *(const bf16x8_t*)(x + m * C + c), r1 = *(const bf16x8_t*)(x + (m + 32) * C + c);
      const bf16x8_t r2 = *(const bf16x8_t*)(x + (m + 64) * C + c), r3 = *(const bf16x8_t*)(x + (m + 96) * C + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += ((float)r0[j] + (float)r1[j]) + ((float)r2[j] + (float)r3[j]);
    }
    for (; m < m1; m += 32) {
      const bf16x8_t r0 = *(const bf16x8_t*)(x + m * C + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)r0[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[pl][8 * v + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64 && c0 + threadIdx.x < C) {
    float t = 0.f;
    for (int q = 0; q < 32; ++q) t += red[q][threadIdx.x];
    part[(size_t)blockIdx.y * C + c0 + threadIdx.x] = t;
  }
}

// ---- ordered sums of partial vectors (dm_sum_item), any number of them in one launch: one workgroup = 64 elements
// of one item x 16 waves; wave w adds the partials w, w + 16, ... in order, the 16 run sums are then added in wave
// order. The item table travels by value in the kernel arguments.
constexpr int OS_WAVES = 16, OS_MAX_ITEMS = 40;
struct OrderedSumGroup {
  dm_sum_item it[OS_MAX_ITEMS];
  int tile0[OS_MAX_ITEMS];   // first 64-element tile of each item in the launch grid
  int count;
};
__global__ __launch_bounds__(64 * OS_WAVES) void ordered_sum_kernel(const OrderedSumGroup g) {
  __shared__ float red[OS_WAVES][64];
  int p = 0;
  while (p + 1 < g.count && g.tile0[p + 1] <= (int)blockIdx.x) ++p;
  const dm_sum_item& a = g.it[p];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int e = (blockIdx.x - g.tile0[p]) * 64 + lane;
  float s = 0.f;
  if (e < a.n) {
    const float* src = a.part + e;
    int q = w;
    for (; q + 3 * OS_WAVES < a.nparts; q += 4 * OS_WAVES) {
      const float v0 = src[(size_t)q * a.stride], v1 = src[(size_t)(q + OS_WAVES) * a.stride];
      const float v2 = src[(size_t)(q + 2 * OS_WAVES) * a.stride], v3 = src[(size_t)(q + 3 * OS_WAVES) * a.stride];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; q < a.nparts; q += OS_WAVES) s += src[(size_t)q * a.stride];
  }
  red[w][lane] = s;
  __syncthreads();
  if (w || e >= a.n) return;
  float t = red[0][lane];
#pragma unroll
  for (int k = 1; k < OS_WAVES; ++k) t += red[k][lane];
  if (e < a.s1) {
    if (a.out0) a.out0[e] += t;
    if (a.dual0) a.dual0[e] += t;
  } else if (e < a.s2) {
    if (a.out1) a.out1[e - a.s1] += t;
  } else if (a.out2) {
    a.out2[e - a.s2] += t;
  }
}

__global__ void colsum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += x[(size_t)r * C + c];
  out[c] += s;
}

// ---- batched [Cout][T][Cin] -> [Cin][T][Cout] transposition of bf16 weights living in one arena ------------
// One workgroup moves one 64x64 (co, ci) tile of one tap through LDS; the item table is device memory so a
// captured graph can replay the launch.
__global__ __launch_bounds__(256) void transpose_weights_kernel(const short* __restrict__ src, short* __restrict__ dst,
                                                                const dm_wt_item* __restrict__ items, int n_items) {
  __shared__ short tile[64][66];
  const int bid = blockIdx.x;
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {  // last item whose tile_begin <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].tile_begin <= bid) lo = mid; else hi = mid - 1;
  }
  const dm_wt_item it = items[lo];
  const int tci = (it.Cin + 63) >> 6, tco = (it.Cout + 63) >> 6;
  int r = bid - it.tile_begin;
  const int t = r / (tco * tci);
  r -= t * tco * tci;
  const int co0 = (r / tci) << 6, ci0 = (r % tci) << 6;
  const short* s = src + it.offset;
  short* d = dst + it.offset;
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int piece = tid + 256 * k, row = piece >> 3, c8 = (piece & 7) << 3;
    const int co = co0 + row, ci = ci0 + c8;
    s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (co < it.Cout && ci < it.Cin) v = *(const s16x8_t*)(s + ((long)co * it.T + t) * it.Cin + ci);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[row][c8 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int piece = tid + 256 * k, row = piece >> 3, c8 = (piece & 7) << 3;  // row: ci, c8: co
    const int ci = ci0 + row, co = co0 + c8;
    if (ci < it.Cin && co < it.Cout) {
      s16x8_t v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[c8 + e][row];
      *(s16x8_t*)(d + ((long)ci * it.T + t) * it.Cout + co) = v;
    }
  }
}

}  // namespace

extern "C" int dm_timestep_embed(const int64_t* t, const float* freq, float* out, int B, int half, void* stream) {
  DM_REQUIRE(t && freq && out && B > 0 && half > 0, DM_ERR_ARG, "dm_timestep_embed: bad argument");
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(dm_cdiv((long)B * half, NT)), dim3(NT), 0, (hipStream_t)stream, t, freq, out, B, half);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act_in,
                             void* stream) {
  DM_REQUIRE(x && w && y && M > 0 && K > 0 && N > 0, DM_ERR_ARG, "dm_linear_fwd: bad argument");
  const int lds = linear_tile_lds_bytes(K);
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(dm_cdiv(N, LIN_COLS), dm_cdiv(M, LIN_ROWS)), dim3(NT), lds, (hipStream_t)stream,
                     x, w, bias, y, M, K, N, act_in);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* dbias, int M,
                             int K, int N, int act_in, void* stream) {
  DM_REQUIRE(x && w && dy && M > 0 && K > 0 && N > 0, DM_ERR_ARG, "dm_linear_bwd: bad argument");
  if (dx) {
    hipLaunchKernelGGL(linear_bwd_x_kernel, dim3(dm_cdiv(K, BX_COLS), dm_cdiv(M, BX_ROWS)), dim3(NT), 0, (hipStream_t)stream,
                       x, w, dy, dx, M, K, N, act_in);
    DM_CHECK_LAUNCH();
  }
  if (dw || dbias) {
    hipLaunchKernelGGL(linear_bwd_w_kernel, dim3(dm_cdiv(K, BW_COLS), dm_cdiv(N, BW_ROWS)), dim3(NT), 0, (hipStream_t)stream,
                       x, dy, dw, dbias, M, K, N, act_in);
    DM_CHECK_LAUNCH();
  }
  return DM_OK;
}

static int fill_group(LinearGroup& g, const dm_linear_item* items, int P, int tile_cols, int* total) {
  DM_REQUIRE(items && P > 0 && P <= MAX_GROUP, DM_ERR_ARG, "dm_linear_group: 1..%d layers per call, got %d", MAX_GROUP, P);
  int t = 0;
  for (int p = 0; p < P; ++p) {
    DM_REQUIRE(items[p].w && items[p].N > 0, DM_ERR_ARG, "dm_linear_group: layer %d has no weight", p);
    g.it[p].w = items[p].w; g.it[p].b = items[p].b; g.it[p].y = items[p].y; g.it[p].dy = items[p].dy;
    g.it[p].dw = items[p].dw; g.it[p].db = items[p].db; g.it[p].N = items[p].N; g.it[p].tile0 = t;
    t += dm_cdiv(items[p].N, tile_cols);
  }
  g.P = P;
  *total = t;
  return DM_OK;
}

extern "C" int dm_linear_group_fwd(const float* x, const dm_linear_item* items, int P, int M, int K, int act_in,
                                   void* stream) {
  DM_REQUIRE(x && M > 0 && K > 0, DM_ERR_ARG, "dm_linear_group_fwd: bad argument");
  LinearGroup g;
  int tiles = 0;
  int rc = fill_group(g, items, P, LIN_COLS, &tiles);
  if (rc) return rc;
  for (int p = 0; p < P; ++p) DM_REQUIRE(items[p].y, DM_ERR_ARG, "dm_linear_group_fwd: layer %d has no output", p);
  const int lds = linear_tile_lds_bytes(K);
  hipLaunchKernelGGL(linear_group_fwd_kernel, dim3(tiles, dm_cdiv(M, LIN_ROWS)), dim3(NT), lds, (hipStream_t)stream, x, g, M, K,
                     act_in);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" long dm_linear_group_bwd_ws_floats(const dm_linear_item* items, int P, int M, int K) {
  if (!items || P <= 0 || M <= 0 || K <= 0) return 0;
  long t = 0;
  for (int p = 0; p < P; ++p) t += dm_cdiv(items[p].N, GT_XN);
  return t * M * K;
}

extern "C" int dm_linear_group_bwd(const float* x, const dm_linear_item* items, int P, float* dx, float* ws, int M, int K,
                                   int act_in, void* stream) {
  DM_REQUIRE(x && M > 0 && K > 0, DM_ERR_ARG, "dm_linear_group_bwd: bad argument");
  DM_REQUIRE(!dx || (ws && (uintptr_t)ws % 16 == 0), DM_ERR_ARG, "dm_linear_group_bwd: dx needs a 16-byte aligned workspace");
  static_assert(NT == 256, "the sum kernel assumes four waves");
  LinearGroup g;
  int tiles = 0;
  int rc = fill_group(g, items, P, GT_N, &tiles);
  if (rc) return rc;
  LinearGroup gx;
  int xtiles = 0;
  rc = fill_group(gx, items, P, GT_XN, &xtiles);
  if (rc) return rc;
  for (int p = 0; p < P; ++p) DM_REQUIRE(items[p].dy, DM_ERR_ARG, "dm_linear_group_bwd: layer %d has no output gradient", p);
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    hipLaunchKernelGGL(linear_group_bwd_x_kernel, dim3(xtiles, dm_cdiv(M, GT_M), dm_cdiv(K, GT_K)), dim3(NT), 0, st, gx, ws, M, K);
    DM_CHECK_LAUNCH();
    hipLaunchKernelGGL(linear_group_bwd_x_sum_kernel, dim3(dm_cdiv((long)M * K, 64)), dim3(NT), 0, st, x, ws, dx, M * K, xtiles, act_in);
    DM_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(linear_group_bwd_w_kernel, dim3(tiles, dm_cdiv(K, GT_K)), dim3(NT), 0, st, x, g, M, K, act_in);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_class_embed_add(float* emb, const float* table, const int64_t* cls, const float* mask, int uncond_idx,
                                  int B, int E, void* stream) {
  DM_REQUIRE(emb && table && B > 0 && E > 0, DM_ERR_ARG, "dm_class_embed_add: bad argument");
  hipLaunchKernelGGL(class_embed_add_kernel, dim3(dm_cdiv((long)B * E, NT)), dim3(NT), 0, (hipStream_t)stream, emb, table, cls, mask, uncond_idx, B, E);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_class_embed_bwd(const float* demb, float* dtable, const int64_t* cls, const float* mask, int uncond_idx,
                                  int n_rows, int B, int E, void* stream) {
  DM_REQUIRE(demb && dtable && B > 0 && E > 0 && n_rows > 0 && n_rows <= 65535, DM_ERR_ARG, "dm_class_embed_bwd: bad argument");
  if (!cls) return DM_OK;  // every row maps to the padding index: no gradient
  hipLaunchKernelGGL(class_embed_bwd_kernel, dim3(dm_cdiv(E, NT), n_rows), dim3(NT), 0, (hipStream_t)stream, demb, dtable, cls, mask, uncond_idx, B, E);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

static int check_resample(const char* who, const void* a, const void* b, int B, int H, int W, int C) {
  DM_REQUIRE(a && b, DM_ERR_ARG, "%s: null argument", who);
  DM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, DM_ERR_SHAPE, "%s: bad shape (C must be a multiple of 8)", who);
  return DM_OK;
}

extern "C" int dm_upsample2x_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  int rc = check_resample("dm_upsample2x_fwd", x, y, B, H, W, C);
  if (rc) return rc;
  const long total = (long)B * 4 * H * W * (C / 8);
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(dm_cdiv(total, NT)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, H, W, C / 8);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
extern "C" int dm_upsample2x_bwd(const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
  int rc = check_resample("dm_upsample2x_bwd", dy, dx, B, H, W, C);
  if (rc) return rc;
  const long total = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(gather2x2_kernel, dim3(dm_cdiv(total, NT)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C / 8, 1.0f);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
extern "C" int dm_avgpool2x_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  int rc = check_resample("dm_avgpool2x_fwd", x, y, B, H, W, C);
  if (rc) return rc;
  DM_REQUIRE(H % 2 == 0 && W % 2 == 0, DM_ERR_SHAPE, "dm_avgpool2x_fwd: odd spatial size %dx%d", H, W);
  const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(gather2x2_kernel, dim3(dm_cdiv(total, NT)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, H / 2, W / 2, C / 8, 0.25f);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
extern "C" int dm_avgpool2x_bwd(const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
  int rc = check_resample("dm_avgpool2x_bwd", dy, dx, B, H, W, C);
  if (rc) return rc;
  DM_REQUIRE(H % 2 == 0 && W % 2 == 0, DM_ERR_SHAPE, "dm_avgpool2x_bwd: odd spatial size %dx%d", H, W);
  const long total = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(scatter2x2_kernel, dim3(dm_cdiv(total, NT)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)dx, B, H / 2, W / 2, C / 8, 0.25f);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_pack_conv_weight(const float* w, long s_co, long s_ci, long s_kh, long s_kw, int Cout, int Cin, int KH,
                                   int KW, void* w_fwd, void* w_dgrad, void* stream) {
  DM_REQUIRE(w && (w_fwd || w_dgrad) && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, DM_ERR_ARG, "dm_pack_conv_weight: bad argument");
  const long total = (long)Cout * Cin * KH * KW;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(dm_cdiv(total, NT)), dim3(NT), 0, (hipStream_t)stream, w, s_co, s_ci, s_kh, s_kw,
                     Cout, Cin, KH, KW, (bf16_t*)w_fwd, (bf16_t*)w_dgrad);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_transpose_conv_weights(const void* src, void* dst, const dm_wt_item* items_dev, int n_items,
                                         int n_tiles, void* stream) {
  DM_REQUIRE(src && dst && items_dev && n_items > 0 && n_tiles > 0, DM_ERR_ARG, "dm_transpose_conv_weights: bad argument");
  DM_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, DM_ERR_ARG, "dm_transpose_conv_weights: arenas must be 16-byte aligned");
  hipLaunchKernelGGL(transpose_weights_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, (const short*)src,
                     (short*)dst, items_dev, n_items);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" int dm_ordered_sum_group(const dm_sum_item* items, int n_items, void* stream) {
  DM_REQUIRE(items && n_items > 0, DM_ERR_ARG, "dm_ordered_sum_group: no items");
  // The kernel adds into its outputs with a plain read-modify-write: two items of ONE launch must not share an
  // output element (a parameter used twice in a forward pass files two sums for the same gradient). Items whose
  // outputs overlap those of an earlier item of the launch being assembled start a new launch behind it, so they
  // add one after the other, in item order.
  struct Range { const float* lo; const float* hi; };
  Range used[4 * OS_MAX_ITEMS];
  int n_used = 0;
  OrderedSumGroup g;
  g.count = 0;
  int tiles = 0;
  auto launch = [&]() -> int {
    if (g.count) {
      hipLaunchKernelGGL(ordered_sum_kernel, dim3(tiles), dim3(64 * OS_WAVES), 0, (hipStream_t)stream, g);
      DM_CHECK_LAUNCH();
    }
    g.count = 0;
    tiles = 0;
    n_used = 0;
    return DM_OK;
  };
  for (int i = 0; i < n_items; ++i) {
    const dm_sum_item& a = items[i];
    DM_REQUIRE(a.part && a.nparts > 0 && a.n > 0 && a.s1 >= 0 && a.s1 <= a.s2 && a.s2 <= a.n && a.stride >= a.n, DM_ERR_ARG,
               "dm_ordered_sum_group: item %d is malformed", i);
    const Range mine[4] = {{a.out0, a.out0 ? a.out0 + a.s1 : nullptr}, {a.dual0, a.dual0 ? a.dual0 + a.s1 : nullptr},
                           {a.out1, a.out1 ? a.out1 + (a.s2 - a.s1) : nullptr}, {a.out2, a.out2 ? a.out2 + (a.n - a.s2) : nullptr}};
    bool clash = false;
    for (int r = 0; r < 4 && !clash; ++r)
      for (int u = 0; u < n_used && !clash; ++u)
        clash = mine[r].lo && mine[r].lo < mine[r].hi && mine[r].lo < used[u].hi && used[u].lo < mine[r].hi;
    if (clash || g.count == OS_MAX_ITEMS) {
      const int rc = launch();
      if (rc != DM_OK) return rc;
    }
    g.it[g.count] = a;
    g.tile0[g.count] = tiles;
    ++g.count;
    tiles += dm_cdiv(a.n, 64);
    for (int r = 0; r < 4; ++r)
      if (mine[r].lo && mine[r].lo < mine[r].hi) used[n_used++] = mine[r];
  }
  return launch();
}

int dm_finish_ordered_sum(const DmOrderedSum& a, dm_sum_item* deferred, hipStream_t st) {
  if (deferred) {
    *deferred = a;
    return DM_OK;
  }
  return dm_ordered_sum_group(&a, 1, (void*)st);
}

// rows per workgroup of the partial-sum kernel: about 1024 workgroups in all, at least 128 rows each
static int colsum_rows_per_block(long M, int C) {
  const int nslab = dm_cdiv(C, 64);
  long chunks = 1024 / nslab > 0 ? 1024 / nslab : 1;
  long rpb = (M + chunks - 1) / chunks;
  rpb = (rpb + 127) / 128 * 128;
  return (int)rpb;
}

extern "C" long dm_colsum_ws_floats(long M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return (long)dm_cdiv(M, colsum_rows_per_block(M, C)) * C;
}

static int colsum_bf16(const void* x, float* out, float* out2, long M, int C, float* ws, dm_sum_item* deferred, void* stream) {
  DM_REQUIRE(x && out && ws && M > 0 && C > 0, DM_ERR_ARG, "dm_colsum_bf16: bad argument");
  DM_REQUIRE(C % 8 == 0, DM_ERR_SHAPE, "dm_colsum_bf16: C=%d must be a multiple of 8", C);
  DM_REQUIRE((uintptr_t)x % 16 == 0, DM_ERR_ARG, "dm_colsum_bf16: misaligned tensor");
  const int rpb = colsum_rows_per_block(M, C), chunks = dm_cdiv(M, rpb);
  DM_REQUIRE(chunks <= 65535, DM_ERR_UNSUPPORTED, "dm_colsum_bf16: %ld rows are too many", M);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_part_kernel, dim3(dm_cdiv(C, 64), chunks), dim3(NT), 0, st, (const bf16_t*)x, ws, M, C, rpb);
  DM_CHECK_LAUNCH();
  DmOrderedSum a = {ws, C, chunks, C, C, C, out, nullptr, nullptr, out2};
  return dm_finish_ordered_sum(a, deferred, st);
}
extern "C" int dm_colsum_bf16(const void* x, float* out, long M, int C, float* ws, dm_sum_item* deferred, void* stream) {
  return colsum_bf16(x, out, nullptr, M, C, ws, deferred, stream);
}
extern "C" int dm_colsum_bf16_dual(const void* x, float* out, float* out2, long M, int C, float* ws, dm_sum_item* deferred,
                                   void* stream) {
  DM_REQUIRE(out2, DM_ERR_ARG, "dm_colsum_bf16_dual: null second output");
  return colsum_bf16(x, out, out2, M, C, ws, deferred, stream);
}
extern "C" int dm_colsum_f32(const float* x, float* out, int R, int C, void* stream) {
  DM_REQUIRE(x && out && R > 0 && C > 0, DM_ERR_ARG, "dm_colsum_f32: bad argument");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3(dm_cdiv(C, NT)), dim3(NT), 0, (hipStream_t)stream, x, out, R, C);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
