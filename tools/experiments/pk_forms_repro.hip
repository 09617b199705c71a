// Which operand-select forms of the packed fp32 instructions lose a product on gfx950 while another kernel's waves run
// matrix-core instructions and direct-to-LDS buffer loads on the same compute units?
//
//   hipcc --offload-arch=gfx950 -O2 -o pk_forms_repro pk_forms_repro.hip && ./pk_forms_repro [launches]
//
// Every thread runs ITER dependent steps of ONE instruction form on operands loaded from global memory; the result is
// compared with the same arithmetic done by unpacked v_fma_f32 / v_mul_f32 / v_add_f32, launch by launch.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
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

constexpr int ITER = 128, THREADS = 256, NFORMS = 14;

struct Form {
  const char* text;
  int op;          // 0 fma, 1 mul (then added to the accumulator unpacked), 2 add
  int sel[3], sel_hi[3];
};
__host__ __device__ constexpr Form form_of(int f) {
  switch (f) {
    case 0: return {"v_pk_fma_f32 (no modifier)", 0, {0, 0, 0}, {1, 1, 1}};
    case 1: return {"v_pk_fma_f32 op_sel:[0,1,0]", 0, {0, 1, 0}, {1, 1, 1}};
    case 2: return {"v_pk_fma_f32 op_sel:[1,0,0]", 0, {1, 0, 0}, {1, 1, 1}};
    case 3: return {"v_pk_fma_f32 op_sel:[0,0,1]", 0, {0, 0, 1}, {1, 1, 1}};
    case 4: return {"v_pk_fma_f32 op_sel:[1,1,0]", 0, {1, 1, 0}, {1, 1, 1}};
    case 5: return {"v_pk_fma_f32 op_sel_hi:[0,1,1]", 0, {0, 0, 0}, {0, 1, 1}};
    case 6: return {"v_pk_fma_f32 op_sel_hi:[1,0,1]", 0, {0, 0, 0}, {1, 0, 1}};
    case 7: return {"v_pk_fma_f32 op_sel_hi:[1,1,0]", 0, {0, 0, 0}, {1, 1, 0}};
    case 8: return {"v_pk_mul_f32 op_sel:[0,1]", 1, {0, 1, 0}, {1, 1, 1}};
    case 9: return {"v_pk_mul_f32 op_sel_hi:[1,0]", 1, {0, 0, 0}, {1, 0, 1}};
    case 10: return {"v_pk_add_f32 op_sel:[0,1]", 2, {0, 1, 0}, {1, 1, 1}};
    case 11: return {"v_pk_add_f32 op_sel_hi:[1,0]", 2, {0, 0, 0}, {1, 0, 1}};
    // v_pk_mov_b32: low = src0[op_sel[0]], high = src1[op_sel[1]] (sel = {source-0 half, -, -}, sel_hi = {-, source-1 half, -})
    case 12: return {"v_pk_mov_b32 op_sel:[1,0]", 3, {1, 0, 0}, {0, 0, 0}};
    default: return {"v_pk_mov_b32 op_sel:[0,1]", 3, {0, 0, 0}, {0, 1, 0}};
  }
}

template <int F>
__device__ __forceinline__ void packed_step(f32x2& acc, f32x2 x, f32x2 w) {
  f32x2 t;
  if (F == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(w));
  if (F == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(x), "v"(w));
  if (F == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(x), "v"(w));
  if (F == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1]" : "+v"(acc) : "v"(x), "v"(w));
  if (F == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0]" : "+v"(acc) : "v"(x), "v"(w));
  if (F == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(x), "v"(w));
  if (F == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(w));
  if (F == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(x), "v"(w));
  if (F == 8) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(x), "v"(w)); acc.x += t.x; acc.y += t.y; }
  if (F == 9) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(w)); acc.x += t.x; acc.y += t.y; }
  if (F == 10) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(x), "v"(w)); acc.x += 0.5f * t.x; acc.y += 0.5f * t.y; }
  if (F == 11) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(w)); acc.x += 0.5f * t.x; acc.y += 0.5f * t.y; }
  if (F == 12) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(x), "v"(w)); acc.x += 0.5f * t.x; acc.y += 0.5f * t.y; }
  if (F == 13) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(x), "v"(w)); acc.x += 0.5f * t.x; acc.y += 0.5f * t.y; }
}

// the same arithmetic, one unpacked instruction per lane of the pair
template <int F>
__device__ __forceinline__ void plain_step(f32x2& acc, f32x2 x, f32x2 w) {
  constexpr Form f = form_of(F);
  const float x_lo = x[f.sel[0]], w_lo = w[f.sel[1]], x_hi = x[f.sel_hi[0]], w_hi = w[f.sel_hi[1]];
  if (f.op == 0) {
    const float c_lo = acc[f.sel[2]], c_hi = acc[f.sel_hi[2]];
    float lo, hi;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(x_lo), "v"(w_lo), "v"(c_lo));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(x_hi), "v"(w_hi), "v"(c_hi));
    acc.x = lo;
    acc.y = hi;
  } else if (f.op == 1) {
    float lo, hi;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(x_lo), "v"(w_lo));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(x_hi), "v"(w_hi));
    acc.x += lo;
    acc.y += hi;
  } else if (f.op == 3) {
    acc.x += 0.5f * x[f.sel[0]];
    acc.y += 0.5f * w[f.sel_hi[1]];
  } else {
    float lo, hi;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(x_lo), "v"(w_lo));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(x_hi), "v"(w_hi));
    acc.x += 0.5f * lo;
    acc.y += 0.5f * hi;
  }
}

template <int F, bool PACKED>
__global__ __launch_bounds__(THREADS) void victim(const float* __restrict__ xin, const float* __restrict__ win,
                                                  float* __restrict__ out) {
  const int t = blockIdx.x * THREADS + threadIdx.x;
  f32x2 acc = {0.25f, -0.5f};
  for (int i = 0; i < ITER; ++i) {
    const f32x2 x = {xin[(i * 2) & 1023], xin[(i * 2 + 1) & 1023]};
    const f32x2 w = {win[(t + i) & 1023], win[(t + i + 7) & 1023]};
    if (PACKED) packed_step<F>(acc, x, w);
    else plain_step<F>(acc, x, w);
  }
  out[2 * t] = acc.x;
  out[2 * t + 1] = acc.y;
}

// the neighbour: matrix-core instructions whose LDS tile is refilled by direct-to-LDS buffer loads every step
__global__ __launch_bounds__(256) void neighbour(float* out, const float* src, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 tile[64 * 1024 / 2];
  for (int i = threadIdx.x; i < 32 * 1024; i += 256) tile[i] = (__bf16)(0.001f * (i & 255));
  __syncthreads();
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x + i)); b[i] = (__bf16)(0.5f - 0.1f * i); }
  f32x16 c0 = {}, c1 = {};
  int off = threadIdx.x * 8;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
  for (int i = 0; i < iters; ++i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(tile + ((i & 7) * 2048 + (threadIdx.x >> 6) * 512)),
                                             16, (threadIdx.x & 63) * 16 + (i & 255) * 4096, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a = *(const bf16x8*)(tile + ((off & 2047) & ~7));
    off += 8;
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int F>
void run_form(int launches, bool with_neighbour, const float* x, const float* w, float* ref, float* out, float* spin,
              const float* src, hipStream_t s1, hipStream_t s2, int blocks) {
  const int n = blocks * THREADS;
  victim<F, false><<<blocks, THREADS, 0, s1>>>(x, w, ref);
  CHECK(hipDeviceSynchronize());
  std::vector<float> href(2 * n), hout((size_t)launches * 2 * n);
  CHECK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemset(out, 0, hout.size() * 4));
  CHECK(hipDeviceSynchronize());
  if (with_neighbour) neighbour<<<2048, 256, 0, s2>>>(spin, src, 5000);
  for (int l = 0; l < launches; ++l) victim<F, true><<<blocks, THREADS, 0, s1>>>(x, w, out + (size_t)l * 2 * n);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(hout.data(), out, hout.size() * 4, hipMemcpyDeviceToHost));
  long bad_launches = 0, lo = 0, hi = 0, lanes[4] = {0, 0, 0, 0};
  for (int l = 0; l < launches; ++l) {
    bool bad = false;
    for (int i = 0; i < 2 * n; ++i)
      if (hout[(size_t)l * 2 * n + i] != href[i]) {
        bad = true;
        (i & 1 ? hi : lo)++;
        lanes[((i / 2) & 63) / 16]++;
      }
    bad_launches += bad;
  }
  printf("%-34s neighbour %d: %4ld of %d launches differ (low lane %ld, high lane %ld; lanes 0-15 %ld, 16-31 %ld, 32-47 %ld, 48-63 %ld)\n",
         form_of(F).text, (int)with_neighbour, bad_launches, launches, lo, hi, lanes[0], lanes[1], lanes[2], lanes[3]);
}

template <int F>
void run_all(int launches, const float* x, const float* w, float* ref, float* out, float* spin, const float* src,
             hipStream_t s1, hipStream_t s2, int blocks) {
  run_form<F>(launches, false, x, w, ref, out, spin, src, s1, s2, blocks);
  run_form<F>(launches, true, x, w, ref, out, spin, src, s1, s2, blocks);
  if constexpr (F + 1 < NFORMS) run_all<F + 1>(launches, x, w, ref, out, spin, src, s1, s2, blocks);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 1000;
  const int blocks = 48, n = blocks * THREADS;
  std::vector<float> hx(1024), hw(1024);
  srand(7);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-3f;
  float *x, *w, *ref, *out, *spin, *src;
  CHECK(hipMalloc(&x, 4096));
  CHECK(hipMalloc(&w, 4096));
  CHECK(hipMalloc(&ref, 2 * n * sizeof(float)));
  CHECK(hipMalloc(&out, (size_t)launches * 2 * n * sizeof(float)));
  CHECK(hipMalloc(&spin, 2048 * 256 * sizeof(float)));
  CHECK(hipMalloc(&src, 2 << 20));
  CHECK(hipMemset(src, 0, 2 << 20));
  CHECK(hipMemcpy(x, hx.data(), 4096, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(w, hw.data(), 4096, hipMemcpyHostToDevice));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  run_all<0>(launches, x, w, ref, out, spin, src, s1, s2, blocks);
  return 0;
}
