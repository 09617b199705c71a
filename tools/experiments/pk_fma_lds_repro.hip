// Stand-alone reproducer attempt, second form: the failing KERNEL SHAPE (an fp32 tile product whose operands come from
// 16-byte LDS reads, FMAs left to the compiler -> v_pk_fma_f32, op_sel:[0,1,0] for the odd elements) beside neighbours
// of three kinds: none, matrix-core instructions only, matrix-core instructions fed from LDS (as a GEMM's main loop is).
//
//   hipcc --offload-arch=gfx950 -O3 -o pk_fma_lds_repro pk_fma_lds_repro.hip && ./pk_fma_lds_repro
//
// victim<0>: compiler FMAs; victim<1>: the same loop with v_fmac_f32 spelled out. Every launch is compared with a quiet
// launch of the v_fmac_f32 form.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

constexpr int NT = 256, ROWS = 16, COLS = 64, M = 8;

// y[m][n] = sum_k x[m][k] w[n][k]: one workgroup = 16 rows x 64 columns, the two tiles staged in LDS
template <int FORM>
__global__ __launch_bounds__(NT) void victim(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                             int N, int K) {
  extern __shared__ float lds[];
  float* xs = lds;                 // [ROWS][K]
  float* ws = lds + ROWS * K;      // [COLS][K + 4]
  const int n0 = blockIdx.x * COLS;
  for (int i = threadIdx.x; i < ROWS * K / 4; i += NT) {
    const int r = i / (K / 4), c4 = i % (K / 4);
    f32x4 v = r < M ? *(const f32x4*)(x + r * K + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    *(f32x4*)(xs + r * K + c4 * 4) = v;
  }
  for (int i = threadIdx.x; i < COLS * K / 4; i += NT) {
    const int c = i / (K / 4), c4 = i % (K / 4);
    *(f32x4*)(ws + c * (K + 4) + c4 * 4) = *(const f32x4*)(w + (size_t)(n0 + c) * K + c4 * 4);
  }
  __syncthreads();
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (K % 4 == 0) {
    for (int k = 0; k < K; k += 4) {
      const f32x4 wv = *(const f32x4*)(ws + c * (K + 4) + k);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 xv = *(const f32x4*)(xs + (rg * 4 + r) * K + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (FORM == 0)
            acc[r] = __builtin_fmaf(xv[e], wv[e], acc[r]);
          else
            asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[r]) : "v"(xv[e]), "v"(wv[e]));
        }
      }
    }
  } else {
    for (int k = 0; k < K; ++k) {
      const float wv = ws[c * (K + 4) + k];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(xs[(rg * 4 + r) * K + k], wv, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = rg * 4 + r;
    if (m < M) y[(size_t)m * N + n0 + c] = acc[r];
  }
}

// neighbours: matrix-core instructions only (0) / operands re-read from LDS every step with ds_read_b128 (1) / with the
// transposing ds_read_b64_tr_b16 (2) / the LDS tile refilled by direct-to-LDS buffer loads every step (3) / the same
// without any matrix-core instruction (4)
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void mfma_spin(float* out, const float* src, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 tile[64 * 1024 / 2];   // 64 KB, the kernel's only LDS: offset 0
  for (int i = threadIdx.x; i < 32 * 1024; i += 256) tile[i] = (__bf16)(0.001f * (i & 255));
  __syncthreads();
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x + i)); b[i] = (__bf16)(0.5f - 0.1f * i); }
  f32x16 c0 = {}, c1 = {};
  int off = threadIdx.x * 8;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
  for (int i = 0; i < iters; ++i) {
    if (KIND == 1) {
      a = *(const bf16x8*)(tile + (off & (32 * 1024 - 8)));
      b = *(const bf16x8*)(tile + ((off + 2048) & (32 * 1024 - 8)));
      off += 2048 + 8;
    }
    if (KIND == 2) {
      s16x4 lo, hi, lo2, hi2;
      const unsigned addr = (unsigned)((off * 2) & (32 * 1024 - 64));
      asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:32\n\t"
                   "ds_read_b64_tr_b16 %2, %4 offset:2048\n\tds_read_b64_tr_b16 %3, %4 offset:2080\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(lo), "=v"(hi), "=v"(lo2), "=v"(hi2) : "v"(addr));
      for (int j = 0; j < 4; ++j) {
        a[j] = __builtin_bit_cast(__bf16, lo[j]); a[4 + j] = __builtin_bit_cast(__bf16, hi[j]);
        b[j] = __builtin_bit_cast(__bf16, lo2[j]); b[4 + j] = __builtin_bit_cast(__bf16, hi2[j]);
      }
      off += 1024 + 8;
    }
    if (KIND == 3 || KIND == 4) {
      // 256 lanes x 16 bytes = 4 KB per step straight into LDS (M0 carries the LDS base of the wave's 1 KB share)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(tile + ((i & 7) * 2048 + (threadIdx.x >> 6) * 512)),
                                               16, (threadIdx.x & 63) * 16 + (i & 255) * 4096, 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      a = *(const bf16x8*)(tile + ((off & 2047) & ~7));
      off += 8;
    }
    if (KIND != 4) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
    } else {
      c0[0] += (float)a[0];
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 300;
  const int tiles = 24, N = tiles * COLS, K = argc > 2 ? atoi(argv[2]) : 64;
  const int lds = (ROWS * K + COLS * (K + 4)) * 4;
  std::vector<float> hx(M * K), hw((size_t)N * K);
  srand(7);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
  float *x, *w, *ref, *out, *spin;
  CHECK(hipMalloc(&x, hx.size() * 4));
  CHECK(hipMalloc(&w, hw.size() * 4));
  CHECK(hipMalloc(&ref, (size_t)M * N * 4));
  CHECK(hipMalloc(&out, (size_t)launches * M * N * 4));
  CHECK(hipMalloc(&spin, 2048 * 256 * 4));
  CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  victim<1><<<tiles, NT, lds, s1>>>(x, w, ref, N, K);
  CHECK(hipStreamSynchronize(s1));
  std::vector<float> href((size_t)M * N), hout((size_t)launches * M * N);
  CHECK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
  int status = 0;
  const char* nb[6] = {"no neighbour", "MFMA only", "MFMA + ds_read_b128", "MFMA + ds_read_b64_tr_b16", "MFMA + buffer_load lds", "buffer_load lds, no MFMA"};
  float* srcbuf;
  CHECK(hipMalloc(&srcbuf, 2 << 20));
  CHECK(hipMemset(srcbuf, 0, 2 << 20));
  const char* fm[2] = {"compiler FMAs (v_pk_fma_f32)", "v_fmac_f32"};
  for (int neighbour = 0; neighbour < 6; ++neighbour)
    for (int form = 0; form < 2; ++form) {
      CHECK(hipMemset(out, 0, hout.size() * 4));
      CHECK(hipDeviceSynchronize());
      if (neighbour == 1) mfma_spin<0><<<2048, 256, 0, s2>>>(spin, srcbuf, 20000);
      if (neighbour == 2) mfma_spin<1><<<2048, 256, 0, s2>>>(spin, srcbuf, 20000);
      if (neighbour == 3) mfma_spin<2><<<2048, 256, 0, s2>>>(spin, srcbuf, 20000);
      if (neighbour == 4) mfma_spin<3><<<2048, 256, 0, s2>>>(spin, srcbuf, 5000);
      if (neighbour == 5) mfma_spin<4><<<2048, 256, 0, s2>>>(spin, srcbuf, 5000);
      for (int l = 0; l < launches; ++l) {
        float* o = out + (size_t)l * M * N;
        if (form == 0) victim<0><<<tiles, NT, lds, s1>>>(x, w, o, N, K);
        else victim<1><<<tiles, NT, lds, s1>>>(x, w, o, N, K);
      }
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(hout.data(), out, hout.size() * 4, hipMemcpyDeviceToHost));
      long bad_launches = 0, bad_elems = 0, last_quarter = 0, even_rows = 0;
      for (int l = 0; l < launches; ++l) {
        bool bad = false;
        for (int i = 0; i < M * N; ++i)
          if (hout[(size_t)l * M * N + i] != href[i]) {
            bad = true;
            ++bad_elems;
            last_quarter += (i % N) % 64 >= 48;
            even_rows += ((i / N) & 1) == 0;
          }
        bad_launches += bad;
      }
      printf("%-26s %-30s: %ld of %d launches differ (%ld elements; in columns 48-63 of a tile: %ld, in even rows: %ld)\n",
             nb[neighbour], fm[form], bad_launches, launches, bad_elems, last_quarter, even_rows);
      if (bad_launches) status = 1;
    }
  return status;
}
