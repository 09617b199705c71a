// Stand-alone check of one instruction form on gfx950: v_pk_fma_f32 with op_sel:[0,1,0] (the low lane multiplies by the
// HIGH register of the second source pair), run while matrix-core (MFMA) waves of another kernel share the compute
// units, against the same arithmetic written as v_mov_b32 + v_pk_fma_f32 op_sel_hi:[1,0,1] and as two v_fmac_f32.
//
//   hipcc --offload-arch=gfx950 -O2 -o pk_fma_opsel_repro pk_fma_opsel_repro.hip && ./pk_fma_opsel_repro
//
// Every thread accumulates acc.lo += x.lo * w.hi, acc.hi += x.hi * w.hi over ITER steps; the three forms must agree bit
// for bit. Prints, per form and with/without the MFMA neighbour, how many launches gave a result that differs from the
// v_fmac_f32 form run alone, and which lanes differed.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                       \
    }                                                                                \
  } while (0)

constexpr int ITER = 256;
constexpr int THREADS = 256;

// form 0: op_sel:[0,1,0]; form 1: v_mov + op_sel_hi:[1,0,1]; form 2: two v_fmac_f32
template <int FORM>
__global__ __launch_bounds__(THREADS) void victim(const float* __restrict__ xin, const float* __restrict__ win,
                                                  float* __restrict__ out) {
  const int t = blockIdx.x * THREADS + threadIdx.x;
  f32x2 acc = {0.f, 0.f}, acc2 = {0.f, 0.f};
  for (int i = 0; i < ITER; ++i) {
    // (the inputs come from memory each step, so that the compiler cannot fold the loop)
    f32x2 x = {xin[(i * 2) & 1023], xin[(i * 2 + 1) & 1023]};
    f32x2 w = {win[(t + i) & 1023], win[(t + i + 7) & 1023]};
    // the sequence of the kernel that failed: two accumulator pairs (rows 0/1 and 2/3), element 0 through the plain
    // form, element 1 through op_sel:[0,1,0], the low operands put in place by v_mov_b32 just before
    f32x2 x2 = {xin[(i * 2 + 512) & 1023], xin[(i * 2 + 513) & 1023]};     // elements 0 and 1 of rows 2 / 3 come next
    f32x2 y = {xin[(i * 2 + 256) & 1023], xin[(i * 2 + 257) & 1023]};
    f32x2 y2 = {xin[(i * 2 + 768) & 1023], xin[(i * 2 + 769) & 1023]};
    if (FORM == 0) {
      asm volatile(
          "v_pk_fma_f32 %0, %2, %6, %0 op_sel_hi:[1,0,1]\n\t"
          "v_pk_fma_f32 %1, %3, %6, %1 op_sel_hi:[1,0,1]\n\t"
          "v_pk_fma_f32 %0, %4, %6, %0 op_sel:[0,1,0]\n\t"
          "v_pk_fma_f32 %1, %5, %6, %1 op_sel:[0,1,0]"
          : "+v"(acc), "+v"(acc2) : "v"(x), "v"(x2), "v"(y), "v"(y2), "v"(w));
    } else if (FORM == 1) {
      float lo;
      asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "v"(w.y));
      f32x2 w2 = {lo, w.y};
      asm volatile(
          "v_pk_fma_f32 %0, %2, %6, %0 op_sel_hi:[1,0,1]\n\t"
          "v_pk_fma_f32 %1, %3, %6, %1 op_sel_hi:[1,0,1]\n\t"
          "v_pk_fma_f32 %0, %4, %7, %0 op_sel_hi:[1,0,1]\n\t"
          "v_pk_fma_f32 %1, %5, %7, %1 op_sel_hi:[1,0,1]"
          : "+v"(acc), "+v"(acc2) : "v"(x), "v"(x2), "v"(y), "v"(y2), "v"(w), "v"(w2));
    } else {
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc.x) : "v"(x.x), "v"(w.x));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc.y) : "v"(x.y), "v"(w.x));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc2.x) : "v"(x2.x), "v"(w.x));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc2.y) : "v"(x2.y), "v"(w.x));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc.x) : "v"(y.x), "v"(w.y));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc.y) : "v"(y.y), "v"(w.y));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc2.x) : "v"(y2.x), "v"(w.y));
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc2.y) : "v"(y2.y), "v"(w.y));
    }
  }
  acc.x += 3.f * acc2.x;
  acc.y += 5.f * acc2.y;
  out[2 * t] = acc.x;
  out[2 * t + 1] = acc.y;
}

// the neighbour: nothing but matrix-core instructions for a few milliseconds, one 256-thread workgroup per launch slot
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_spin_bf16(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x + i)); b[i] = (__bf16)(0.5f - 0.1f * i); }
  f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ __launch_bounds__(256) void mfma_spin(float* out, int iters) {
  f16x4 a = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)(0.001f * threadIdx.x)};
  f16x4 b = {(_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f, (_Float16)0.125f};
  f32x16 c0 = {}, c1 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x8f16(b, a, c1, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200;
  const int blocks = 64;
  const int n = blocks * THREADS;
  std::vector<float> hx(1024), hw(1024);
  srand(7);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-3f;
  float *x, *w, *ref, *out, *spin;
  CHECK(hipMalloc(&x, 4096));
  CHECK(hipMalloc(&w, 4096));
  CHECK(hipMalloc(&ref, 2 * n * sizeof(float)));
  CHECK(hipMalloc(&out, (size_t)launches * 2 * n * sizeof(float)));
  CHECK(hipMalloc(&spin, 1024 * 256 * sizeof(float)));
  CHECK(hipMemcpy(x, hx.data(), 4096, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(w, hw.data(), 4096, hipMemcpyHostToDevice));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));

  victim<2><<<blocks, THREADS, 0, s1>>>(x, w, ref);
  CHECK(hipStreamSynchronize(s1));
  std::vector<float> href(2 * n), hout((size_t)launches * 2 * n);
  CHECK(hipMemcpy(href.data(), ref, 2 * n * sizeof(float), hipMemcpyDeviceToHost));

  int status = 0;
  for (int neighbour = 0; neighbour < 3; ++neighbour) {
    for (int form = 0; form < 3; ++form) {
      CHECK(hipMemset(out, 0, (size_t)launches * 2 * n * sizeof(float)));
      CHECK(hipDeviceSynchronize());
      if (neighbour == 1) mfma_spin<<<1024, 256, 0, s2>>>(spin, 20000);   // a few milliseconds on every compute unit
      if (neighbour == 2) mfma_spin_bf16<<<1024, 256, 0, s2>>>(spin, 40000);
      for (int l = 0; l < launches; ++l) {
        float* o = out + (size_t)l * 2 * n;
        if (form == 0) victim<0><<<blocks, THREADS, 0, s1>>>(x, w, o);
        if (form == 1) victim<1><<<blocks, THREADS, 0, s1>>>(x, w, o);
        if (form == 2) victim<2><<<blocks, THREADS, 0, s1>>>(x, w, o);
      }
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(hout.data(), out, (size_t)launches * 2 * n * sizeof(float), hipMemcpyDeviceToHost));
      long bad_launches = 0, bad_lo = 0, bad_hi = 0, lanes[4] = {0, 0, 0, 0};
      for (int l = 0; l < launches; ++l) {
        bool bad = false;
        for (int i = 0; i < 2 * n; ++i)
          if (hout[(size_t)l * 2 * n + i] != href[i]) {
            bad = true;
            (i & 1 ? bad_hi : bad_lo)++;
            lanes[((i / 2) & 63) / 16]++;
          }
        bad_launches += bad;
      }
      const char* names[3] = {"v_pk_fma_f32 op_sel:[0,1,0]", "v_mov_b32 + v_pk_fma_f32 op_sel_hi:[1,0,1]", "2 x v_fmac_f32"};
      printf("MFMA neighbour %d, %-44s: %ld of %d launches differ (low-lane results %ld, high-lane %ld; lanes 0-15 %ld, 16-31 %ld, 32-47 %ld, 48-63 %ld)\n",
             neighbour, names[form], bad_launches, launches, bad_lo, bad_hi, lanes[0], lanes[1], lanes[2], lanes[3]);
      if (bad_launches) status = 1;
    }
  }
  return status;
}
