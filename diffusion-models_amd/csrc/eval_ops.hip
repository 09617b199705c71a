// Kernels either side of the training/sampling path:
//   - the data path: a batch gathered from a uint8 image set that lives in HBM, flipped and normalised on the device
//     (replaces the DataLoader + torchvision transform pipeline of train.py:109-138, 312-329 for data that fits HBM)
//   - the memorisation metric: for every generated image the two nearest training images in L2 distance
//     (utils/validate_memo.py:5-28: torch.cdist + topk(2)), as an fp32 MFMA GEMM with the top-2 selection fused in
#include "dm_common.h"

namespace {

constexpr int NT = 256;

// ---- uint8 [N][H][W][C] -> fp32 [B][C][H][W], rows chosen by idx, optional flips, normalised to [-1, 1]
//   mode 0: ((v / 255) - 0.5) / 0.5   ToTensor + Normalize(0.5, 0.5), the transform of the training DataLoader (train.py:111-114)
//   mode 1: v / 127.5 - 1             the data_tensor form (train.py:124-125, :136)
// every operation is a separately rounded fp32 operation, as torch evaluates them (this file is built without FMA
// contraction)
__global__ void u8_batch_kernel(const uint8_t* __restrict__ data, const int64_t* __restrict__ idx,
                                const uint8_t* __restrict__ flip, float* __restrict__ out, int B, int C, int H, int W,
                                long n_items, int mode) {
  const long total = (long)B * C * H * W;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const int w = i % W;
    const int h = (i / W) % H;
    const int c = (i / ((long)W * H)) % C;
    const int b = i / ((long)W * H * C);
    long item = idx ? idx[b] : b;
    if (item < 0 || item >= n_items) { out[i] = 0.f; continue; }   // host checks the indices; never read out of range
    const int f = flip ? flip[b] : 0;
    const int ws = (f & 1) ? W - 1 - w : w;
    const int hs = (f & 2) ? H - 1 - h : h;
    const float v = (float)data[((item * H + hs) * W + ws) * C + c];
    float r;
    if (mode == 0) r = __fdiv_rn(__fsub_rn(__fdiv_rn(v, 255.0f), 0.5f), 0.5f);
    else r = __fsub_rn(__fdiv_rn(v, 127.5f), 1.0f);
    out[i] = r;
  }
}

// ---- squared row norms, fp32, one wave per row (fixed order)
__global__ __launch_bounds__(NT) void row_sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, long n, int D) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* p = x + row * D;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) s += p[k] * p[k];
  s = dm_wave_sum(s);
  if (lane == 0) out[row] = s;
}

// ---- two nearest reference rows of every generated row.
// d2[j][i] = |ref_j|^2 + |gen_i|^2 - 2 ref_j . gen_i, the dot products on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: bf16 inputs would lose the small distances of memorised samples to cancellation).
// One workgroup = 64 generated rows x one slice of the reference rows, walked in tiles of 64; 4 waves as 2 (ref) x 2
// (gen), each a 32 x 32 accumulator whose lanes hold ONE generated row (column) and 16 reference rows, so the running
// top-2 of a generated row is lane-local. Tiles go through LDS in K-chunks of 32 floats.
constexpr int NN_T = 64, NN_K = 32, NN_PITCH = NN_K + 1;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ void top2_push(float& b1, float& b2, float v) {
  const float lo = fminf(b1, v), hi = fmaxf(b1, v);
  b1 = lo;
  b2 = fminf(b2, hi);
}

__global__ __launch_bounds__(NT) void nn2_kernel(const float* __restrict__ gen, const float* __restrict__ ref,
                                                 const float* __restrict__ gnorm, const float* __restrict__ rnorm,
                                                 float* __restrict__ part, int n_gen, int n_ref, int D, int refs_per_split) {
  __shared__ float sg[NN_T * NN_PITCH], sr[NN_T * NN_PITCH];
  __shared__ float sb[2][NN_T][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wg = wave & 1;             // wave's 32 reference rows / 32 generated rows of the tile
  const int g0 = blockIdx.x * NN_T;
  const int r_begin = blockIdx.y * refs_per_split, r_end = min(n_ref, r_begin + refs_per_split);
  const int col = lane & 31, half = lane >> 5;
  const int gi = g0 + wg * 32 + col;                   // this lane's generated row
  const float gn = gi < n_gen ? gnorm[gi] : 0.f;
  float b1 = INFINITY, b2 = INFINITY;
  for (int r0 = r_begin; r0 < r_end; r0 += NN_T) {
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < D; k0 += NN_K) {
      __syncthreads();
      // 64 rows x 32 floats of each operand: 2048 floats = 8 per thread, row-major with a padded pitch
      for (int e = tid; e < NN_T * NN_K; e += NT) {
        const int row = e / NN_K, k = e - row * NN_K;
        const bool kok = k0 + k < D;
        sg[row * NN_PITCH + k] = (g0 + row < n_gen && kok) ? gen[(size_t)(g0 + row) * D + k0 + k] : 0.f;
        sr[row * NN_PITCH + k] = (r0 + row < r_end && kok) ? ref[(size_t)(r0 + row) * D + k0 + k] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < NN_K; kk += 2) {
        // A (rows = reference): lane holds A[col][kk + half]; B (columns = generated): lane holds B[kk + half][col]
        const float av = sr[(wr * 32 + col) * NN_PITCH + kk + half];
        const float bv = sg[(wg * 32 + col) * NN_PITCH + kk + half];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
    }
    // accumulator register r of this lane: reference row 8 * (r / 4) + 4 * half + (r % 4), generated column col
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rj = r0 + wr * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
      if (rj < r_end) top2_push(b1, b2, fmaxf(rnorm[rj] + gn - 2.f * acc[r], 0.f));
    }
  }
  // merge the two half-waves (same generated row, other reference rows), then the two reference halves of the tile
  {
    const float o1 = __shfl_xor(b1, 32, 64), o2 = __shfl_xor(b2, 32, 64);
    const float m1 = fminf(b1, o1), m2 = fminf(fmaxf(b1, o1), fminf(b2, o2));
    b1 = m1; b2 = m2;
  }
  __syncthreads();
  if (half == 0) { sb[wr][wg * 32 + col][0] = b1; sb[wr][wg * 32 + col][1] = b2; }
  __syncthreads();
  if (tid < NN_T && g0 + tid < n_gen) {
    const float a1 = sb[0][tid][0], a2 = sb[0][tid][1], c1 = sb[1][tid][0], c2 = sb[1][tid][1];
    float* o = part + ((size_t)(g0 + tid) * gridDim.y + blockIdx.y) * 2;
    o[0] = fminf(a1, c1);
    o[1] = fminf(fmaxf(a1, c1), fminf(a2, c2));
  }
}

// final merge over the reference slices; distances (not squared) out
__global__ void nn2_merge_kernel(const float* __restrict__ part, float* __restrict__ d1, float* __restrict__ d2, int n_gen,
                                 int splits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_gen) return;
  float b1 = INFINITY, b2 = INFINITY;
  for (int s = 0; s < splits; ++s) {
    top2_push(b1, b2, part[((size_t)i * splits + s) * 2]);
    top2_push(b1, b2, part[((size_t)i * splits + s) * 2 + 1]);
  }
  d1[i] = sqrtf(b1);
  d2[i] = sqrtf(b2);
}

int nn2_splits(int n_gen, int n_ref) {
  const int row_tiles = dm_cdiv(n_gen, NN_T), ref_tiles = dm_cdiv(n_ref, NN_T);
  int s = dm_cdiv(1024, row_tiles);          // about four workgroups per CU
  if (s > ref_tiles) s = ref_tiles;
  return s < 1 ? 1 : s;
}

}  // namespace

extern "C" int dm_u8_batch(const uint8_t* data, const int64_t* idx, const uint8_t* flip, float* out, int B, int C, int H,
                           int W, long n_items, int mode, void* stream) {
  DM_REQUIRE(data && out && B > 0 && C > 0 && H > 0 && W > 0 && n_items > 0 && (mode == 0 || mode == 1), DM_ERR_ARG,
             "dm_u8_batch: bad argument");
  const long total = (long)B * C * H * W;
  long g = (total + NT - 1) / NT;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(u8_batch_kernel, dim3((int)g), dim3(NT), 0, (hipStream_t)stream, data, idx, flip, out, B, C, H, W,
                     n_items, mode);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

extern "C" long dm_nn2_ws_floats(int n_gen, int n_ref) {
  if (n_gen <= 0 || n_ref <= 0) return 0;
  return (long)n_gen + n_ref + 2L * n_gen * nn2_splits(n_gen, n_ref);
}

extern "C" int dm_nn2_distances(const float* gen, const float* ref, float* d1, float* d2, int n_gen, int n_ref, int D,
                                float* ws, void* stream) {
  DM_REQUIRE(gen && ref && d1 && d2 && ws && n_gen > 0 && n_ref >= 2 && D > 0, DM_ERR_ARG,
             "dm_nn2_distances: bad argument (at least two reference rows)");
  hipStream_t st = (hipStream_t)stream;
  float* gnorm = ws;
  float* rnorm = ws + n_gen;
  float* part = rnorm + n_ref;
  hipLaunchKernelGGL(row_sumsq_kernel, dim3(dm_cdiv(n_gen, NT / 64)), dim3(NT), 0, st, gen, gnorm, (long)n_gen, D);
  DM_CHECK_LAUNCH();
  hipLaunchKernelGGL(row_sumsq_kernel, dim3(dm_cdiv(n_ref, NT / 64)), dim3(NT), 0, st, ref, rnorm, (long)n_ref, D);
  DM_CHECK_LAUNCH();
  const int splits = nn2_splits(n_gen, n_ref);
  const int per = dm_cdiv(dm_cdiv(n_ref, splits), NN_T) * NN_T;
  const int used = dm_cdiv(n_ref, per);
  DM_REQUIRE(used <= splits && used <= 65535, DM_ERR_UNSUPPORTED, "dm_nn2_distances: too many reference slices");
  hipLaunchKernelGGL(nn2_kernel, dim3(dm_cdiv(n_gen, NN_T), used), dim3(NT), 0, st, gen, ref, gnorm, rnorm, part, n_gen,
                     n_ref, D, per);
  DM_CHECK_LAUNCH();
  hipLaunchKernelGGL(nn2_merge_kernel, dim3(dm_cdiv(n_gen, NT)), dim3(NT), 0, st, part, d1, d2, n_gen, used);
  DM_CHECK_LAUNCH();
  return DM_OK;
}
