// Probe of the gfx950 primitives the conv/wgrad kernels are built on.
// Each probe fills known integer data, runs one workgroup, and compares with
// the lane mapping the kernels assume.  Prints PASS/FAIL per probe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

// ---- probe 1: global_load_lds 16 B: dest = wave-uniform base + lane*16, source per lane
__global__ void p1(const short* g, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[256 * 8 + 64];
  int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int src = (tid ^ 5);
  __builtin_amdgcn_global_load_lds(GLB_PTR(g + src * 8), LDS_PTR(lds + w * 512), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 8; ++j) out[tid * 8 + j] = lds[w * 512 + lane * 8 + j];
}

// ---- probe 2: ds_read_b64_tr_b16
__global__ void p2(short* out) {
  __shared__ __attribute__((aligned(16))) short M[64 * 64];
  int tid = threadIdx.x;
  for (int i = tid; i < 64 * 64; i += 64) M[i] = (short)i;
  __syncthreads();
  int g = tid >> 4, i = tid & 15, q = i >> 2, p = i & 3;
  const short* addr = &M[(8 * g + q) * 64 + 16 + 4 * p];
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[tid * 4 + e] = r[e];
}

// ---- probe 3: mfma 16x16x32 bf16 lane maps
__global__ void p3(const short* A, const short* B, float* D) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// ---- probe 4: buffer_load ... lds with out-of-range lanes: does LDS receive zeros?
__global__ void p4(const short* g, short* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 8];
  int tid = threadIdx.x;
  for (int j = 0; j < 8; ++j) lds[tid * 8 + j] = 0x7777;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  unsigned voff = (tid & 1) ? 0x80000000u : (unsigned)(tid * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 8; ++j) out[tid * 8 + j] = lds[tid * 8 + j];
}

// ---- probe 5: mfma 32x32x16 bf16 lane maps
__global__ void p5(const short* A, const short* B, float* D) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j]; b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)]; }
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

int main() {
  int fails = 0;
  { // p1
    std::vector<short> g(256 * 8); for (size_t i = 0; i < g.size(); ++i) g[i] = (short)(i * 3 + 1);
    short *dg, *dout; CK(hipMalloc(&dg, g.size() * 2)); CK(hipMalloc(&dout, g.size() * 2));
    CK(hipMemcpy(dg, g.data(), g.size() * 2, hipMemcpyHostToDevice));
    p1<<<1, 256>>>(dg, dout); CK(hipDeviceSynchronize());
    std::vector<short> o(g.size()); CK(hipMemcpy(o.data(), dout, o.size() * 2, hipMemcpyDeviceToHost));
    int bad = 0; for (int t = 0; t < 256; ++t) for (int j = 0; j < 8; ++j) if (o[t * 8 + j] != g[(t ^ 5) * 8 + j]) ++bad;
    printf("probe1 global_load_lds lane-linear dest: %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    if (bad) { for (int t = 0; t < 8; ++t) printf(" t%d got %d exp %d\n", t, o[t * 8], g[(t ^ 5) * 8]); }
  }
  { // p2
    short* dout; CK(hipMalloc(&dout, 64 * 4 * 2));
    p2<<<1, 64>>>(dout); CK(hipDeviceSynchronize());
    std::vector<short> o(256); CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < 64; ++t) for (int e = 0; e < 4; ++e) { int g = t >> 4, i = t & 15; int exp = (8 * g + e) * 64 + 16 + i; if (o[t * 4 + e] != exp) ++bad; }
    printf("probe2 ds_read_b64_tr_b16 map: %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    if (bad) for (int t = 0; t < 64; ++t) printf(" lane%d: %d %d %d %d  (row,col of e0: %d,%d)\n", t, o[t*4], o[t*4+1], o[t*4+2], o[t*4+3], o[t*4] / 64, o[t*4] % 64);
  }
  { // p3
    std::vector<short> A(16 * 32), B(32 * 16); std::vector<float> Af(16 * 32), Bf(32 * 16);
    for (int i = 0; i < 16 * 32; ++i) { Af[i] = (float)((i * 7 + 3) % 13 - 6); A[i] = (short)f2bf(Af[i]); }
    for (int i = 0; i < 32 * 16; ++i) { Bf[i] = (float)((i * 5 + 1) % 11 - 5); B[i] = (short)f2bf(Bf[i]); }
    short *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    p3<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    std::vector<float> D(256); CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += Af[i * 32 + k] * Bf[k * 16 + j]; if (s != D[i * 16 + j]) ++bad; }
    printf("probe3 mfma_f32_16x16x32_bf16 maps: %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  { // p4
    std::vector<short> g(64 * 8); for (size_t i = 0; i < g.size(); ++i) g[i] = (short)(i + 1);
    short *dg, *dout; CK(hipMalloc(&dg, g.size() * 2)); CK(hipMalloc(&dout, g.size() * 2));
    CK(hipMemcpy(dg, g.data(), g.size() * 2, hipMemcpyHostToDevice));
    p4<<<1, 64>>>(dg, dout, (int)(g.size() * 2)); CK(hipDeviceSynchronize());
    std::vector<short> o(g.size()); CK(hipMemcpy(o.data(), dout, o.size() * 2, hipMemcpyDeviceToHost));
    int bad_in = 0, zero_oob = 0, stale_oob = 0, other = 0;
    for (int t = 0; t < 64; ++t) for (int j = 0; j < 8; ++j) {
      short v = o[t * 8 + j];
      if (t & 1) { if (v == 0) ++zero_oob; else if (v == 0x7777) ++stale_oob; else ++other; }
      else if (v != g[t * 8 + j]) ++bad_in;
    }
    printf("probe4 buffer_load_lds: in-range bad=%d; OOB lanes: zero=%d stale=%d other=%d  => %s\n", bad_in, zero_oob, stale_oob, other,
           (bad_in == 0 && zero_oob == 32 * 8) ? "PASS(zero-fill)" : "INFO");
  }
  { // p5
    std::vector<short> A(32 * 16), B(16 * 32); std::vector<float> Af(32 * 16), Bf(16 * 32);
    for (int i = 0; i < 32 * 16; ++i) { Af[i] = (float)((i * 7 + 3) % 13 - 6); A[i] = (short)f2bf(Af[i]); }
    for (int i = 0; i < 16 * 32; ++i) { Bf[i] = (float)((i * 5 + 1) % 11 - 5); B[i] = (short)f2bf(Bf[i]); }
    short *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    p5<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    std::vector<float> D(1024); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += Af[i * 16 + k] * Bf[k * 32 + j]; if (s != D[i * 32 + j]) ++bad; }
    printf("probe5 mfma_f32_32x32x16_bf16 maps: %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device: %s CUs=%d clock=%d MHz L2=%d\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1000, pr.l2CacheSize);
  return fails ? 1 : 0;
}
