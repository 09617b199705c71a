// Weight gradient of the implicit-GEMM convolution on bf16 MFMA (gfx950).
//
//   dw[n][tap][c] += sum_m dy[m][n] * X(m, tap)[c]
//
// The reduction index m (output pixel) is the SLOW index of both operands in memory (NHWC), so both MFMA
// operands are column slices of row-major [pixel][channel] LDS tiles. The tiles are filled by
// buffer_load ... lds (zero rows for padding / rows beyond M) and the fragments are read with
// ds_read_b64_tr_b16, the gfx950 transposing LDS read: two reads give a lane its 8 consecutive pixels of one
// channel. 32-byte units of each row are XOR-swizzled (source side) so the 8 rows one half-wave touches per
// transposing read land on different banks.
//
// One workgroup = one (tap, 128|64 output channels, 128|64 input channels) tile over one slice of the pixel
// axis (split-K). dw ACCUMULATES (callers zero it or keep the running gradient in it).
//
// Float atomics execute at the memory side at about 1.3 TB/s chip-wide, a fifth of the plain-store rate, and one
// round of workgroups adds a whole CU-full of accumulators (25 MB): every kernel therefore stores its partial tile
// as a slab in a caller-provided workspace (plain 16-byte stores) and a second, light kernel sums the slabs of a
// tile in slice order and adds the sum to dw - which also makes the result independent of arrival order (bitwise
// reproducible). There is no atomic path.
#include "dm_common.h"

extern int g_dm_use_wgrad3;
extern int g_dm_use_wgrad9;
extern int g_dm_wgrad_slab_overhead;
extern int g_dm_wgrad_reduce_cap;
extern int g_dm_timing_skip;
extern int g_dm_timing_twice;

namespace {

constexpr int BKM = 64;  // pixels per step

struct WgradArgs {
  const void* x1;
  const void* x2;
  const void* dy;
  float* dw;
  float* slabs;   // partial tiles [workgroup][SLAB_FLOATS] (3-tap kernel with a workspace), else null
  int C1, C2, Cin, Cout;
  int B, Hs, Ws, Hi, Wi, Ho, Wo, KH, KW, stride, pad, ups;
  int M, splitk, rows_per_split, n_co_tiles, n_ci_tiles, n_blocks;
  unsigned x1_bytes, x2_bytes, dy_bytes;
  unsigned magic_hw, shift_hw, magic_w, shift_w;
  int db_step, dh_step, dw_step;  // 64 pixels as (images, rows, columns) of the output
};

__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned magic, unsigned shift) {
  return (__umulhi(n, magic) + n) >> shift;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// swizzle of the 16-byte piece index within a row, by row; PPR = pieces per row (16: 256-byte rows, 8: 128-byte)
template <int PPR>
__device__ __forceinline__ int swz(int row) {
  if (PPR == 16) return (((row & 3) | ((row >> 1) & 4)) << 1);
  return ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1);
}

template <int TW, int NT>  // tile width in channels (64 or 128), threads per workgroup
struct Tile {
  static constexpr int PPR = TW / 8;
  static constexpr int ROW_BYTES = TW * 2;
  static constexpr int BYTES = BKM * ROW_BYTES;
  static constexpr int PIECES = BKM * PPR / NT;  // per thread
  static constexpr int ROW_STEP = NT / PPR;
};

// Per-lane byte offset (within a tile) of the first transposing read of the fragment whose 16 channels start at
// colbase; the second read of the fragment is 4 rows further, the fragment of the next 32 pixels 32 rows further
// (both keep the swizzle term, which only looks at row bits 0, 1 and 3).
template <int TW>
__device__ __forceinline__ int frag_offset(int colbase, int lane) {
  constexpr int PPR = TW / 8, ROW_BYTES = TW * 2;
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int piece = (colbase >> 3) + (p >> 1);
  const int r0 = 8 * g + q;
  return r0 * ROW_BYTES + ((piece ^ swz<PPR>(r0)) << 4) + (p & 1) * 8;
}

// Two ds_read_b64_tr_b16 give a lane its 8 consecutive pixels of one channel. Issued as inline asm so that the
// compiler does not drain the LDS-DMA ring (vmcnt(0)) in front of them; the caller waits with frag_wait().
template <int OFF0, int OFF1>
__device__ __forceinline__ void read_frag(s16x4_t& lo, s16x4_t& hi, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi)
               : "v"(addr), "i"(OFF0), "i"(OFF1));
}

__device__ __forceinline__ bf16x8_t join_frag(s16x4_t lo, s16x4_t hi) {
  s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

// Wait for every outstanding LDS read; naming the destinations read-write keeps their consumers below the wait.
__device__ __forceinline__ void frag_wait(s16x4_t (&a)[4], s16x4_t (&b)[4], s16x4_t (&c)[4], s16x4_t (&d)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                 "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
}
__device__ __forceinline__ void frag_wait(s16x4_t (&a)[4], s16x4_t (&b)[4], s16x4_t (&c)[2], s16x4_t (&d)[2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                 "+v"(c[0]), "+v"(c[1]), "+v"(d[0]), "+v"(d[1]));
}
__device__ __forceinline__ void frag_wait(s16x4_t (&a)[2], s16x4_t (&b)[2], s16x4_t (&c)[4], s16x4_t (&d)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]),
                 "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
}
__device__ __forceinline__ void frag_wait(s16x4_t (&a)[2], s16x4_t (&b)[2], s16x4_t (&c)[2], s16x4_t (&d)[2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(c[0]), "+v"(c[1]), "+v"(d[0]), "+v"(d[1]));
}

// One workgroup per CU: KG groups of 4 waves; group g multiplies pixels [32g, 32g+32) of every 64-pixel step (KG = 2)
// or all 64 (KG = 1), so each SIMD holds KG waves that hide each other's LDS latency. The tiles move through a ring
// of NST LDS stages filled by buffer_load ... lds NST-1 steps ahead; a step waits only for its own stage (counted
// vmcnt; steps beyond the end still issue their loads, out of range = zero fill, to keep the count fixed).
template <int TCO, int TCI, int NST, int KG>
__global__ __launch_bounds__(256 * KG, 1) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int NT = 256 * KG;
  using TY = Tile<TCO, NT>;
  using TX = Tile<TCI, NT>;
  constexpr int STAGE = TY::BYTES + TX::BYTES;
  constexpr int LPS = TY::PIECES + TX::PIECES;  // loads per thread per stage
  constexpr int EP_PITCH = TCI + 4;
  constexpr int EP_BYTES = TCO * EP_PITCH * 4;
  constexpr int TOTAL = (NST * STAGE > EP_BYTES) ? NST * STAGE : EP_BYTES;
  constexpr int MI = TCO / 32, NJ = TCI / 32;  // 16x16 tiles per wave (waves 2x2 per group)
  static_assert(LPS * (NST - 2) < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) char smem[TOTAL];

  // XCD x runs the contiguous range [x*n/8, (x+1)*n/8) of logical ids, ordered pixel-slice major, so the taps and
  // tiles of one pixel slice (which re-read the same rows) sit on one XCD's L2
  const int per = a.KH * a.KW * a.n_co_tiles * a.n_ci_tiles;
  const int logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (logical >= a.n_blocks) return;
  const int ks = logical / per;
  const int inner = logical - ks * per;
  const int tiles = a.n_co_tiles * a.n_ci_tiles;
  const int tap = inner / tiles;
  const int rest = inner - tap * tiles;
  const int tco = rest / a.n_ci_tiles, tci = rest - tco * a.n_ci_tiles;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int co0 = tco * TCO, ci0 = tci * TCI;
  const bool second = ci0 >= a.C1;
  const int csrc = second ? a.C2 : a.C1;
  const int cioff = second ? ci0 - a.C1 : ci0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;

  __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(second ? a.x2 : a.x1), 0,
                                                                 second ? a.x2_bytes : a.x1_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);

  const int m_begin = ks * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);
  const int nsteps = (m_end - m_begin + BKM - 1) / BKM;

  // staging geometry
  const int yrow = tid / TY::PPR, yslot = tid % TY::PPR;
  const int ypiece = yslot ^ swz<TY::PPR>(yrow);
  const int xrow = tid / TX::PPR, xslot = tid % TX::PPR;
  const int xpiece = xslot ^ swz<TX::PPR>(xrow);
  // output channels beyond Cout (partial tile) read zeros
  const bool ycol_ok = co0 + ypiece * 8 < a.Cout;
  const bool xcol_ok = cioff + xpiece * 8 < csrc;

  // Row cursors. A thread stages the same row slots in every step and consecutive steps are 64 pixels apart, so
  // the (image, row, column) of a slot is decoded once with divisions and then advanced by a fixed (db, dh, dw)
  // with one carry each; addresses need only full-rate 24-bit multiply-adds (pixel counts and row pitches are far
  // below 2^24 and the byte offsets below 2^31).
  int cb[TX::PIECES], coh[TX::PIECES], cow[TX::PIECES];  // image * Hs, output row, output column
  const int sshift = a.stride - 1, hs_step = a.db_step * a.Hs;
#pragma unroll
  for (int i = 0; i < TX::PIECES; ++i) {
    const unsigned m = (unsigned)(m_begin + xrow + TX::ROW_STEP * i);
    const unsigned b = fastdiv(m, a.magic_hw, a.shift_hw);
    const unsigned r = m - b * (unsigned)(a.Ho * a.Wo);
    const unsigned oh = fastdiv(r, a.magic_w, a.shift_w);
    cb[i] = (int)b * a.Hs; coh[i] = (int)oh; cow[i] = (int)(r - oh * (unsigned)a.Wo);
  }
  unsigned yoff[TY::PIECES];
#pragma unroll
  for (int i = 0; i < TY::PIECES; ++i)
    yoff[i] = (unsigned)(((m_begin + yrow + TY::ROW_STEP * i) * a.Cout + co0 + ypiece * 8) * 2);
  const unsigned ystep = (unsigned)(BKM * a.Cout * 2);
  const int xconst = (cioff + xpiece * 8) * 2, csrc2 = csrc * 2;
  int mb = m_begin;  // first pixel of the next step to stage

  auto stage = [&](int step) {  // called with step = 0, 1, 2, ... in order
    char* sy = smem + (step % NST) * STAGE;
    char* sx = sy + TY::BYTES;
#pragma unroll
    for (int i = 0; i < TY::PIECES; ++i) {
      const int m = mb + yrow + TY::ROW_STEP * i;
      const unsigned off = (m < m_end && ycol_ok) ? yoff[i] : DM_OOB;
      yoff[i] += ystep;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, DM_LDS(sy + (wave * 64 + NT * i) * 16), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TX::PIECES; ++i) {
      const int m = mb + xrow + TX::ROW_STEP * i;
      int ih = (coh[i] << sshift) - a.pad + kh, iw = (cow[i] << sshift) - a.pad + kw;   // stride is 1 or 2
      const bool ok = m < m_end && xcol_ok && ih >= 0 && iw >= 0 && ih < a.Hi && iw < a.Wi;
      ih >>= a.ups; iw >>= a.ups;
      const unsigned pix = (unsigned)__umul24((unsigned)(cb[i] + ih), (unsigned)a.Ws) + iw;
      const unsigned off = ok ? (unsigned)__umul24(pix, (unsigned)csrc2) + xconst : DM_OOB;
      // advance the cursor by one step
      cow[i] += a.dw_step;
      const int c1 = cow[i] >= a.Wo;
      cow[i] -= c1 ? a.Wo : 0;
      coh[i] += a.dh_step + c1;
      const int c2 = coh[i] >= a.Ho;
      coh[i] -= c2 ? a.Ho : 0;
      cb[i] += hs_step + (c2 ? a.Hs : 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, DM_LDS(sx + (wave * 64 + NT * i) * 16), 16, off, 0, 0, 0);
    }
    mb += BKM;
  };

  // fragment read addresses within a stage
  const unsigned lds_base = (unsigned)(size_t)DM_LDS(smem);
  unsigned a_off[MI], b_off[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
    a_off[i] = frag_offset<TCO>(wm * (TCO / 2) + i * 16, lane) + (KG == 2 ? kg * 32 * TY::ROW_BYTES : 0);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    b_off[j] = TY::BYTES + frag_offset<TCI>(wn * (TCI / 2) + j * 16, lane) + (KG == 2 ? kg * 32 * TX::ROW_BYTES : 0);

  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) stage(s);
  for (int s = 0; s < nsteps; ++s) {
    // this step's stage has landed for this wave (NST-2 younger stages may still be in flight) ...
    wait_vmcnt<LPS*(NST - 2)>();
    // ... and for every wave; every wave is also done reading the stage of step s-1, which is refilled next
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned sbase = lds_base + (unsigned)((s % NST) * STAGE);
    auto compute = [&]() {
#pragma unroll
      for (int kq = 0; kq < 2 / KG; ++kq) {
        // KG == 2: the wave group's half of the step is folded into a_off/b_off; KG == 1: both halves by immediate
        s16x4_t alo[MI], ahi[MI], blo[NJ], bhi[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (kq == 0) read_frag<0, 4 * TY::ROW_BYTES>(alo[i], ahi[i], sbase + a_off[i]);
          else read_frag<32 * TY::ROW_BYTES, 36 * TY::ROW_BYTES>(alo[i], ahi[i], sbase + a_off[i]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (kq == 0) read_frag<0, 4 * TX::ROW_BYTES>(blo[j], bhi[j], sbase + b_off[j]);
          else read_frag<32 * TX::ROW_BYTES, 36 * TX::ROW_BYTES>(blo[j], bhi[j], sbase + b_off[j]);
        }
        frag_wait(alo, ahi, blo, bhi);
        bf16x8_t af[MI], bfr[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = join_frag(alo[i], ahi[i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = join_frag(blo[j], bhi[j]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    };
    stage(s + NST - 1);
    compute();
  }

  // ---- epilogue: sum the wave groups and transpose through LDS; the partial tile leaves row-major with plain
  // 16-byte stores (wgrad_reduce_kernel adds the slabs of a tile up)
  wait_vmcnt<0>();  // the trailing zero-fill loads must have landed before LDS is reused
  __syncthreads();
  float* ep = (float*)smem;
  const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    if (kg == g) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* e = ep + (wm * (TCO / 2) + i * 16 + fq * 4 + r) * EP_PITCH + wn * (TCI / 2) + j * 16 + frow;
            *e = (g == 0) ? acc[i][j][r] : *e + acc[i][j][r];
          }
    }
    __syncthreads();
  }
  f32x4_t* slab = (f32x4_t*)a.slabs + (size_t)logical * (TCO * TCI / 4);
  for (int item = tid; item < TCO * TCI / 4; item += NT) {
    const int row = item / (TCI / 4), c4 = item - row * (TCI / 4);
    slab[item] = *(const f32x4_t*)(ep + row * EP_PITCH + c4 * 4);
  }
}

// Sum of the row-major TCO x TCI slabs of one (tap, tile) over the pixel slices, added to dw: a wave owns 64
// consecutive float4s of the tile, G waves of a workgroup share them (wave g sums slices g, g + G, ...; fixed order).
template <int TCO, int TCI>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs a, int G) {
  __shared__ f32x4_t red[4][64];
  constexpr int F4 = TCO * TCI / 4, CH = F4 / 64;
  const int T = a.KH * a.KW;
  const int per = T * a.n_co_tiles * a.n_ci_tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int units = per * CH * G;
  for (int u0 = blockIdx.x * 4; u0 < units; u0 += gridDim.x * 4) {
    const int unit = u0 + wave;
    const int cid = unit / G, grp = unit - cid * G;
    const bool live = cid < per * CH;
    const int inner = live ? cid / CH : 0, chunk = live ? cid - inner * CH : 0;
    const int e = chunk * 64 + lane;
    const f32x4_t* src = (const f32x4_t*)a.slabs + (size_t)inner * F4 + e;
    const size_t stride = (size_t)per * F4;
    f32x4_t sum = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (live) {
      int ks = grp;
      for (; ks + 3 * G < a.splitk; ks += 4 * G) {
        const f32x4_t v0 = src[(size_t)ks * stride], v1 = src[(size_t)(ks + G) * stride];
        const f32x4_t v2 = src[(size_t)(ks + 2 * G) * stride], v3 = src[(size_t)(ks + 3 * G) * stride];
        sum += v0; sum += v1; sum += v2; sum += v3;
      }
      for (; ks < a.splitk; ks += G) sum += src[(size_t)ks * stride];
    }
    if (G > 1) {
      __syncthreads();
      red[wave][lane] = sum;
      __syncthreads();
      if (grp == 0)
        for (int g = 1; g < G; ++g) sum += red[wave + g][lane];
    }
    if (!live || grp) continue;
    const int tiles = a.n_co_tiles * a.n_ci_tiles;
    const int tap = inner / tiles, rest = inner - tap * tiles;
    const int tco = rest / a.n_ci_tiles, tci = rest - tco * a.n_ci_tiles;
    const int row = e / (TCI / 4), c4 = e - row * (TCI / 4);
    const int co = tco * TCO + row, ci = tci * TCI + c4 * 4;
    if (co < a.Cout) {
      f32x4_t* dst = (f32x4_t*)(a.dw + ((size_t)co * T + tap) * a.Cin + ci);
      *dst = *dst + sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 layers whose output width divides 64: one workgroup owns a 128 (co) x 64 (ci) tile for the
// THREE taps of one kernel row. A 64-pixel step is RS = 64 / Wo whole output rows; the input rows they need are
// staged once WITH their left/right halo ([RS][Wo + 2] pixels) and the three taps read them at column offsets
// 0, 1, 2, so a step moves ~26 KB into LDS for 3 x as many MFMAs as the single-tap kernel's 24 KB (the L2 -> LDS
// path, not the matrix core, bounds these kernels). The dy fragments are shared by the three taps as well.
// Same ring / wave-group / counted-vmcnt structure as conv_wgrad_kernel.
template <int N>
__device__ __forceinline__ void frag_wait_n(s16x4_t (&a)[N], s16x4_t (&b)[N]);
template <>
__device__ __forceinline__ void frag_wait_n<4>(s16x4_t (&a)[4], s16x4_t (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ void frag_wait6(s16x4_t (&a)[6], s16x4_t (&b)[6]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]),
                 "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]));
}
__device__ __forceinline__ void read_tr(s16x4_t& v, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
}

constexpr int SLAB_FLOATS = 3 * 128 * 64;   // one partial tile of the 3-tap kernel

template <int NST>
__device__ __forceinline__ void wgrad3_body(const WgradArgs& a, const int logical) {
  constexpr int NT = 512, TCO = 128, TCI = 64;
  using TY = Tile<TCO, NT>;                       // dy tile: 64 pixels x 128 channels
  constexpr int XBYTES = 16384;                   // patch region: up to 128 pixels x 64 channels
  constexpr int STAGE = TY::BYTES + XBYTES;
  constexpr int XPIECES = XBYTES / 16 / NT;       // 2 per thread
  constexpr int LPS = TY::PIECES + XPIECES;
  constexpr int EP_BYTES = 24 * 256 * 16;         // epilogue: one wave group's accumulators (24 float4 per thread)
  constexpr int TOTAL = (NST * STAGE > EP_BYTES) ? NST * STAGE : EP_BYTES;
  constexpr int MI = 4, NJ = 2;                   // wave tile 64 (co) x 32 (ci)
  __shared__ __attribute__((aligned(16))) char smem[TOTAL];

  const int per = 3 * a.n_co_tiles * a.n_ci_tiles;
  const int ks = logical / per;
  const int inner = logical - ks * per;
  const int tiles = a.n_co_tiles * a.n_ci_tiles;
  const int kh = inner / tiles;
  const int rest = inner - kh * tiles;
  const int tco = rest / a.n_ci_tiles, tci = rest - tco * a.n_ci_tiles;
  const int co0 = tco * TCO, ci0 = tci * TCI;
  const bool second = ci0 >= a.C1;
  const int csrc = second ? a.C2 : a.C1;
  const int cioff = second ? ci0 - a.C1 : ci0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;

  __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(second ? a.x2 : a.x1), 0,
                                                                 second ? a.x2_bytes : a.x1_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);

  const int m_begin = ks * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);
  const int nsteps = (m_end - m_begin + BKM - 1) / BKM;
  const int PW = a.Wo + 2, RS = BKM / a.Wo, npatch = RS * PW;

  // Swizzle of the 16-byte pieces of patch pixel pp = (prow, pcol). One transposing read serves 4 + 4 pixels that are
  // 8 apart in the 64-pixel step: neighbours in one patch row on images >= 16 wide (the pixel-index key of the plain
  // tiles separates them), but the SAME columns of two consecutive patch rows on 8-wide images, where the row pitch
  // of 10 makes that key collide - there the key is (column bit 1, row bit 0).
  const bool narrow = a.Wo == 8;
  auto patch_swz = [&](int pp, int prow, int pcol) {
    return narrow ? ((((pcol >> 1) & 1) | ((prow & 1) << 1)) << 1) : swz<8>(pp);
  };
  // dy staging (as in the single-tap kernel)
  const int yrow = tid / TY::PPR, yslot = tid % TY::PPR;
  const int ypiece = yslot ^ swz<TY::PPR>(yrow);
  const bool ycol_ok = co0 + ypiece * 8 < a.Cout;
  unsigned yoff[TY::PIECES];
#pragma unroll
  for (int i = 0; i < TY::PIECES; ++i)
    yoff[i] = (unsigned)(((m_begin + yrow + TY::ROW_STEP * i) * a.Cout + co0 + ypiece * 8) * 2);
  const unsigned ystep = (unsigned)(BKM * a.Cout * 2);

  // patch staging: piece idx = tid + 512 i -> patch pixel idx / 8 (row prow, column pcol incl. halo), 16-byte slot idx % 8
  int crow[XPIECES], cbh[XPIECES];      // cursor: output row within its image, image * Hs
  int xcol[XPIECES], xconst[XPIECES];   // source column (>> ups) or -1, channel byte offset
  const int gr0 = m_begin / a.Wo;       // first (stacked) output row of the slice
#pragma unroll
  for (int i = 0; i < XPIECES; ++i) {
    const int idx = tid + NT * i, pp = idx >> 3, slot = idx & 7;
    const int prow = pp / PW, pcol = pp - prow * PW;
    const int gr = gr0 + prow;
    const int b = gr / a.Ho;
    cbh[i] = b * a.Hs;
    crow[i] = gr - b * a.Ho;
    const int iw = pcol - 1;
    xcol[i] = (pp < npatch && iw >= 0 && iw < a.Wi) ? (iw >> a.ups) : -1;
    xconst[i] = (cioff + ((slot ^ patch_swz(pp, prow, pcol)) << 3)) * 2;
  }
  const int db_hs = (RS / a.Ho) * a.Hs, dh = RS % a.Ho;
  const int csrc2 = csrc * 2;
  const int rows_end = (m_end + a.Wo - 1) / a.Wo;  // stacked output rows of this slice end here
  int grow = gr0;                                  // stacked output row of patch row 0 of the next step to stage

  auto stage = [&](int step) {  // called with step = 0, 1, 2, ... in order
    char* sy = smem + (step % NST) * STAGE;
    char* sx = sy + TY::BYTES;
    const int mb = m_begin + step * BKM;
#pragma unroll
    for (int i = 0; i < TY::PIECES; ++i) {
      const int m = mb + yrow + TY::ROW_STEP * i;
      const unsigned off = (m < m_end && ycol_ok) ? yoff[i] : DM_OOB;
      yoff[i] += ystep;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, DM_LDS(sy + (wave * 64 + NT * i) * 16), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < XPIECES; ++i) {
      const int pp = (tid + NT * i) >> 3;
      const int prow = pp / PW;
      int ih = crow[i] - 1 + kh;
      const bool ok = xcol[i] >= 0 && grow + prow < rows_end && ih >= 0 && ih < a.Hi;
      ih >>= a.ups;
      const unsigned pix = (unsigned)__umul24((unsigned)(cbh[i] + ih), (unsigned)a.Ws) + xcol[i];
      const unsigned off = ok ? (unsigned)__umul24(pix, (unsigned)csrc2) + xconst[i] : DM_OOB;
      crow[i] += dh;
      const int c = crow[i] >= a.Ho;
      crow[i] -= c ? a.Ho : 0;
      cbh[i] += db_hs + (c ? a.Hs : 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, DM_LDS(sx + (wave * 64 + NT * i) * 16), 16, off, 0, 0, 0);
    }
    grow += RS;
  };

  // fragment read addresses within a stage
  const unsigned lds_base = (unsigned)(size_t)DM_LDS(smem);
  unsigned a_off[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) a_off[i] = frag_offset<TCO>(wm * 64 + i * 16, lane) + kg * 32 * TY::ROW_BYTES;
  // patch fragments: lane's two pixel rows (r0, r0 + 4) of this wave group's 32 pixels, per tap, per 16-channel tile
  unsigned b_off[3][2][NJ];
  {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = kg * 32 + 8 * g + q + 4 * h;
      const int prow = r / a.Wo, pcol = r - prow * a.Wo;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int pp = prow * PW + pcol + t;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int piece = ((wn * 32 + j * 16) >> 3) + (p >> 1);
          b_off[t][h][j] = TY::BYTES + pp * 128 + ((piece ^ patch_swz(pp, prow, pcol + t)) << 4) + (p & 1) * 8;
        }
      }
    }
  }

  f32x4_t acc[3][MI][NJ];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) stage(s);
  // The two waves of a SIMD (one per wave group) run half a step out of phase: group 0 reads its fragments and
  // multiplies them in the same step, group 1 multiplies the fragments it read in the previous step first and then
  // reads the next ones, so one wave's LDS latency and address arithmetic sit under the other wave's MFMAs.
  s16x4_t alo[MI], ahi[MI], blo[6], bhi[6];
  auto read_frags = [&](int s) {
    const unsigned sbase = lds_base + (unsigned)((s % NST) * STAGE);
#pragma unroll
    for (int i = 0; i < MI; ++i) read_frag<0, 4 * TY::ROW_BYTES>(alo[i], ahi[i], sbase + a_off[i]);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        read_tr(blo[t * 2 + j], sbase + b_off[t][0][j]);
        read_tr(bhi[t * 2 + j], sbase + b_off[t][1][j]);
      }
  };
  auto multiply = [&]() {
    frag_wait_n<4>(alo, ahi);
    frag_wait6(blo, bhi);
    bf16x8_t af[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = join_frag(alo[i], ahi[i]);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bf16x8_t bfr = join_frag(blo[t * 2 + j], bhi[t * 2 + j]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[t][i][j], 0, 0, 0);
      }
  };
  const bool lead = __builtin_amdgcn_readfirstlane(kg) == 0;
  for (int s = 0; s < nsteps; ++s) {
    wait_vmcnt<LPS*(NST - 2)>();
    // group 1's fragment reads of the previous step must have left LDS before its stage is refilled (next barrier on)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!lead && s > 0) multiply();
    read_frags(s);
    stage(s + NST - 1);
    if (lead) multiply();
  }
  if (!lead && nsteps > 0) multiply();

  wait_vmcnt<0>();
  __syncthreads();
  // ---- epilogue: sum the two wave groups through LDS in accumulator-register order (float4 q = (tap, i, j) of
  // thread t at [q][t]: conflict-free 16-byte accesses) and store the slab with plain stores
  f32x4_t* ep4 = (f32x4_t*)smem;
  const int t2 = tid & 255;
  if (kg == 1) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) ep4[((t * MI + i) * NJ + j) * 256 + t2] = acc[t][i][j];
  }
  __syncthreads();
  if (kg == 0) {
    f32x4_t* slab = (f32x4_t*)a.slabs + (size_t)logical * (SLAB_FLOATS / 4);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int q = (t * MI + i) * NJ + j;
          slab[q * 256 + t2] = ep4[q * 256 + t2] + acc[t][i][j];
        }
  }
}

// Workgroup b of a grid runs on XCD b % 8: logical block indices are dealt so that each XCD owns a contiguous run
// (consecutive logical blocks share a pixel slice, i.e. the same dy / input rows, in that XCD's L2).
__device__ __forceinline__ int xcd_logical_block() {
  return (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
}

template <int NST>
__global__ __launch_bounds__(512, 1) void conv_wgrad3_kernel(WgradArgs a) {
  const int logical = xcd_logical_block();
  if (logical >= a.n_blocks) return;
  wgrad3_body<NST>(a, logical);
}

// ------------------------------------------------------------------------------------------------------------
// All NINE taps of a 3x3 / stride 1 / pad 1 layer in one workgroup, on a 128 (co) x 32 (ci) tile. A 64-pixel step is
// RS = 64 / Wo whole output rows of ONE image; the RS + 2 input rows they need are staged once with their left/right
// halo and serve every tap, so a step moves the bytes of the three-tap kernel's step (dy 16 KB + <= 9 KB of input)
// for 1.5 x its MFMAs, and the dy fragments are shared by nine taps instead of three. Patch pixels are 64-byte rows
// (32 channels = two 32-byte halves); the halves of pixel pp swap places when bit 3 of pp is set (8-wide images: when
// the patch row is odd), so the 4 + 4 pixel rows that half a wave touches in one transposing read fall on different
// banks whatever the tap. Fragment reads run one kernel row ahead of the MFMAs inside the step.
constexpr int SLAB9_FLOATS = 9 * 128 * 32;

template <int N>
__device__ __forceinline__ void wait9(s16x4_t (&a)[4], s16x4_t (&b)[4], s16x4_t (&c)[3], s16x4_t (&d)[3]) {
  asm volatile("s_waitcnt lgkmcnt(%14)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                 "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2])
               : "n"(N));
}

template <int NST>
__device__ __forceinline__ void wgrad9_body(const WgradArgs& a, const int logical) {
  constexpr int NT = 512, TCO = 128, TCI = 32;
  using TY = Tile<TCO, NT>;                       // dy tile: 64 pixels x 128 channels
  constexpr int XBYTES = 16384;                   // patch region: every thread issues two 16-byte pieces; <= 9 KB used
  constexpr int STAGE = TY::BYTES + XBYTES;
  constexpr int XPIECES = 2;
  constexpr int LPS = TY::PIECES + XPIECES;
  constexpr int EP_TAPS = 5;                      // epilogue: one wave group's accumulators, five taps at a time, so
  constexpr int EP_BYTES = EP_TAPS * 4 * 256 * 16;   // that the ring decides the LDS footprint (what is left of the
  constexpr int TOTAL = (NST * STAGE > EP_BYTES) ? NST * STAGE : EP_BYTES;   // CU's LDS takes the HBM-bound kernels
  constexpr int MI = 4;                           // of the other stream); wave tile 64 (co) x 16 (ci)
  __shared__ __attribute__((aligned(16))) char smem[TOTAL];

  const int per = a.n_co_tiles * a.n_ci_tiles;
  const int ks = logical / per;
  const int inner = logical - ks * per;
  const int tco = inner / a.n_ci_tiles, tci = inner - tco * a.n_ci_tiles;
  const int co0 = tco * TCO, ci0 = tci * TCI;
  const bool second = ci0 >= a.C1;
  const int csrc = second ? a.C2 : a.C1;
  const int cioff = second ? ci0 - a.C1 : ci0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;

  __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(second ? a.x2 : a.x1), 0,
                                                                 second ? a.x2_bytes : a.x1_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);

  const int m_begin = ks * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);
  const int nsteps = (m_end - m_begin + BKM - 1) / BKM;
  const int PW = a.Wo + 2, RS = BKM / a.Wo, npatch = (RS + 2) * PW;
  const bool narrow = a.Wo == 8;
  auto half_key = [&](int pp, int prow) { return narrow ? ((prow & 1) << 1) : (((pp >> 3) & 1) << 1); };

  // dy staging (as in the other kernels)
  const int yrow = tid / TY::PPR, yslot = tid % TY::PPR;
  const int ypiece = yslot ^ swz<TY::PPR>(yrow);
  const bool ycol_ok = co0 + ypiece * 8 < a.Cout;
  unsigned yoff[TY::PIECES];
#pragma unroll
  for (int i = 0; i < TY::PIECES; ++i)
    yoff[i] = (unsigned)(((m_begin + yrow + TY::ROW_STEP * i) * a.Cout + co0 + ypiece * 8) * 2);
  const unsigned ystep = (unsigned)(BKM * a.Cout * 2);

  // patch staging: piece idx = tid + 512 i -> patch pixel idx / 4 (row prow incl. the two halo rows, column pcol incl.
  // halo), 16-byte slot idx % 4
  int xprow[XPIECES], xcol[XPIECES], xconst[XPIECES];
#pragma unroll
  for (int i = 0; i < XPIECES; ++i) {
    const int idx = tid + NT * i, pp = idx >> 2, slot = idx & 3;
    const int prow = pp / PW, pcol = pp - prow * PW;
    const int iw = pcol - 1;
    xprow[i] = prow;
    xcol[i] = (pp < npatch && iw >= 0 && iw < a.Wi) ? (iw >> a.ups) : -1;
    xconst[i] = (cioff + ((slot ^ half_key(pp, prow)) << 3)) * 2;
  }
  const int csrc2 = csrc * 2;
  // the step's first output row: image and row within it (a step never leaves its image: Ho is a multiple of RS)
  int img = (m_begin / a.Wo) / a.Ho;
  int row0 = m_begin / a.Wo - img * a.Ho;

  auto stage = [&](int step) {  // called with step = 0, 1, 2, ... in order
    char* sy = smem + (step % NST) * STAGE;
    char* sx = sy + TY::BYTES;
    const int mb = m_begin + step * BKM;
#pragma unroll
    for (int i = 0; i < TY::PIECES; ++i) {
      const int m = mb + yrow + TY::ROW_STEP * i;
      const unsigned off = (m < m_end && ycol_ok) ? yoff[i] : DM_OOB;
      yoff[i] += ystep;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, DM_LDS(sy + (wave * 64 + NT * i) * 16), 16, off, 0, 0, 0);
    }
    const bool live = mb < m_end;
#pragma unroll
    for (int i = 0; i < XPIECES; ++i) {
      const int ih = row0 - 1 + xprow[i];
      const bool ok = live && xcol[i] >= 0 && ih >= 0 && ih < a.Hi;
      const unsigned pix = (unsigned)__umul24((unsigned)(img * a.Hs + (ih >> a.ups)), (unsigned)a.Ws) + xcol[i];
      const unsigned off = ok ? (unsigned)__umul24(pix, (unsigned)csrc2) + xconst[i] : DM_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, DM_LDS(sx + (wave * 64 + NT * i) * 16), 16, off, 0, 0, 0);
    }
    row0 += RS;
    if (row0 >= a.Ho) { row0 = 0; ++img; }
  };

  // fragment read addresses within a stage
  const unsigned lds_base = (unsigned)(size_t)DM_LDS(smem);
  unsigned a_off[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) a_off[i] = frag_offset<TCO>(wm * 64 + i * 16, lane) + kg * 32 * TY::ROW_BYTES;
  // patch fragments: lane's two pixel rows (r0, r0 + 4) of this wave group's 32 pixels, per tap
  unsigned b_off[3][3][2];
  {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = kg * 32 + 8 * g + q + 4 * h;
      const int prow = r / a.Wo, pcol = r - prow * a.Wo;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int pp = (prow + kh) * PW + pcol + kw;
          const int piece = 2 * wn + (p >> 1);
          b_off[kh][kw][h] = TY::BYTES + pp * 64 + ((piece ^ half_key(pp, prow + kh)) << 4) + (p & 1) * 8;
        }
    }
  }

  f32x4_t acc[9][MI];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) stage(s);
  s16x4_t alo[MI], ahi[MI], blo[2][3], bhi[2][3];
  for (int s = 0; s < nsteps; ++s) {
    wait_vmcnt<LPS*(NST - 2)>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned sbase = lds_base + (unsigned)((s % NST) * STAGE);
#pragma unroll
    for (int i = 0; i < MI; ++i) read_frag<0, 4 * TY::ROW_BYTES>(alo[i], ahi[i], sbase + a_off[i]);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      read_tr(blo[0][kw], sbase + b_off[0][kw][0]);
      read_tr(bhi[0][kw], sbase + b_off[0][kw][1]);
    }
    stage(s + NST - 1);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      read_tr(blo[1][kw], sbase + b_off[1][kw][0]);
      read_tr(bhi[1][kw], sbase + b_off[1][kw][1]);
    }
    wait9<6>(alo, ahi, blo[0], bhi[0]);        // the six reads of kernel row 1 may still be in flight
    bf16x8_t af[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = join_frag(alo[i], ahi[i]);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const bf16x8_t bfr = join_frag(blo[0][kw], bhi[0][kw]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
        acc[kw][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[kw][i], 0, 0, 0);
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {           // kernel row 2 into the registers kernel row 0 has left
      read_tr(blo[0][kw], sbase + b_off[2][kw][0]);
      read_tr(bhi[0][kw], sbase + b_off[2][kw][1]);
    }
    wait9<6>(alo, ahi, blo[1], bhi[1]);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const bf16x8_t bfr = join_frag(blo[1][kw], bhi[1][kw]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
        acc[3 + kw][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[3 + kw][i], 0, 0, 0);
    }
    wait9<0>(alo, ahi, blo[0], bhi[0]);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const bf16x8_t bfr = join_frag(blo[0][kw], bhi[0][kw]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
        acc[6 + kw][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[6 + kw][i], 0, 0, 0);
    }
  }

  wait_vmcnt<0>();
  __syncthreads();
  // ---- epilogue: sum the two wave groups through LDS in accumulator-register order (float4 q = (tap, i) of thread t
  // at [q][t]) and store the slab with plain stores
  f32x4_t* ep4 = (f32x4_t*)smem;
  const int t2 = tid & 255;
  f32x4_t* slab = (f32x4_t*)a.slabs + (size_t)logical * (SLAB9_FLOATS / 4);
#pragma unroll
  for (int t0 = 0; t0 < 9; t0 += EP_TAPS) {
    if (t0) __syncthreads();   // the first batch has been read
    if (kg == 1) {
#pragma unroll
      for (int t = t0; t < t0 + EP_TAPS && t < 9; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i) ep4[((t - t0) * MI + i) * 256 + t2] = acc[t][i];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int t = t0; t < t0 + EP_TAPS && t < 9; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
          slab[(t * MI + i) * 256 + t2] = ep4[((t - t0) * MI + i) * 256 + t2] + acc[t][i];
    }
  }
}

template <int NST>
__global__ __launch_bounds__(512, 1) void conv_wgrad9_kernel(WgradArgs a) {
  const int logical = xcd_logical_block();
  if (logical >= a.n_blocks) return;
  wgrad9_body<NST>(a, logical);
}

// Slab sum of the nine-tap kernel (same scheme as wgrad3_reduce_round; float4 q = (tap, i) of thread t2).
__global__ __launch_bounds__(256) void wgrad9_reduce_kernel(WgradArgs a, int G) {
  __shared__ f32x4_t red[4][64];
  constexpr int F4 = SLAB9_FLOATS / 4, CH = F4 / 64;
  const int per = a.n_co_tiles * a.n_ci_tiles;
  const int units = per * CH * G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int u0 = blockIdx.x * 4; u0 < units; u0 += gridDim.x * 4) {
    const int unit = u0 + wave;
    const int cid = unit / G, grp = unit - cid * G;
    const bool live = cid < per * CH;
    const int inner = live ? cid / CH : 0, chunk = live ? cid - inner * CH : 0;
    const int e = chunk * 64 + lane;
    const f32x4_t* src = (const f32x4_t*)a.slabs + (size_t)inner * F4 + e;
    const size_t stride = (size_t)per * F4;
    f32x4_t sum = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (live) {
      int ks = grp;
      for (; ks + 3 * G < a.splitk; ks += 4 * G) {
        const f32x4_t v0 = src[(size_t)ks * stride], v1 = src[(size_t)(ks + G) * stride];
        const f32x4_t v2 = src[(size_t)(ks + 2 * G) * stride], v3 = src[(size_t)(ks + 3 * G) * stride];
        sum += v0; sum += v1; sum += v2; sum += v3;
      }
      for (; ks < a.splitk; ks += G) sum += src[(size_t)ks * stride];
    }
    if (G > 1) {
      __syncthreads();
      red[wave][lane] = sum;
      __syncthreads();
      if (grp == 0)
        for (int g = 1; g < G; ++g) sum += red[wave + g][lane];
    }
    if (live && grp == 0) {
      const int q = e >> 8, t2 = e & 255;
      const int tco = inner / a.n_ci_tiles, tci = inner - tco * a.n_ci_tiles;
      const int tap = q >> 2, i = q & 3;
      const int pw = t2 >> 6, l2 = t2 & 63, wm = pw >> 1, wn = pw & 1;
      const int co = tco * 128 + wm * 64 + i * 16 + (l2 >> 4) * 4;
      const int ci = tci * 32 + wn * 16 + (l2 & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (co + r < a.Cout) a.dw[((size_t)(co + r) * 9 + tap) * a.Cin + ci] += sum[r];
    }
  }
}

// Several layers in one launch: the weight gradients of a backward pass feed nothing but the optimizer, so a few
// consecutive layers can wait for each other and share the chip. Every launch leaves one partial tile per workgroup
// (a CU-full of accumulators, 25 MB) whatever the layer, and rounds its workgroup count up to whole rounds of CUs;
// with G layers per launch each layer is cut into 1/G as many pixel slices, so G times fewer partial tiles are
// written and summed, the slices are G times longer (fixed costs amortise) and the rounding loss is shared.
constexpr int WG_MAX = 8;
struct WgradGroup {
  WgradArgs it[WG_MAX];
  int start[WG_MAX + 1];    // first logical block of each layer; start[n] = all blocks
  int ustart[WG_MAX + 1];   // first unit (wave-sized work item of the slab sum) of each layer, multiples of 4
  int G[WG_MAX];            // waves sharing a chunk of the slab sum, per layer
  int n;
};

template <int NST>
__global__ __launch_bounds__(512, 1) void conv_wgrad3_group_kernel(WgradGroup g) {
  const int logical = xcd_logical_block();
  if (logical >= g.start[g.n]) return;
  int l = 0;
  while (l + 1 < g.n && logical >= g.start[l + 1]) ++l;
  const WgradArgs a = g.it[l];
  wgrad3_body<NST>(a, logical - g.start[l]);
}

// Sums the slabs of one tile over the pixel slices and adds the sum to dw, in a fixed order. A wave owns 64
// consecutive float4s of a tile's slab (a slice is 1 KiB per wave load); G = 1, 2 or 4 waves of a workgroup share such
// a chunk when there are enough slices - wave g sums slices g, g + G, ... and the partial sums are added in wave
// order. Lane t of the 64 owns the float4 that thread (q, t2) of the producing workgroups' first wave group held:
// 4 consecutive output channels of one input channel.
// One round of a workgroup: its four waves take units u0 .. u0 + 3 of layer `a` (G the same for all four).
__device__ __forceinline__ void wgrad3_reduce_round(const WgradArgs& a, const int G, const int u0, f32x4_t (*red)[64]) {
  constexpr int F4 = SLAB_FLOATS / 4, CH = F4 / 64;   // float4s per slab, 64-float4 chunks per slab
  const int per = 3 * a.n_co_tiles * a.n_ci_tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const int unit = u0 + wave;
    const int cid = unit / G, grp = unit - cid * G;      // chunk (over all tiles), slice group
    const bool live = cid < per * CH;
    const int inner = live ? cid / CH : 0, chunk = live ? cid - inner * CH : 0;
    const int e = chunk * 64 + lane;                     // float4 index in the slab = q * 256 + t2
    const f32x4_t* src = (const f32x4_t*)a.slabs + (size_t)inner * F4 + e;
    const size_t stride = (size_t)per * F4;
    f32x4_t sum = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (live) {
      int ks = grp;
      for (; ks + 3 * G < a.splitk; ks += 4 * G) {
        const f32x4_t v0 = src[(size_t)ks * stride], v1 = src[(size_t)(ks + G) * stride];
        const f32x4_t v2 = src[(size_t)(ks + 2 * G) * stride], v3 = src[(size_t)(ks + 3 * G) * stride];
        sum += v0; sum += v1; sum += v2; sum += v3;
      }
      for (; ks < a.splitk; ks += G) sum += src[(size_t)ks * stride];
    }
    if (G > 1) {
      __syncthreads();   // the previous round's partial sums have been consumed
      red[wave][lane] = sum;
      __syncthreads();
      if (grp == 0)
        for (int g = 1; g < G; ++g) sum += red[wave + g][lane];
    }
    if (!live || grp) return;

    const int q = e >> 8, t2 = e & 255;
    const int tiles = a.n_co_tiles * a.n_ci_tiles;
    const int kh = inner / tiles, rest = inner - kh * tiles;
    const int tco = rest / a.n_ci_tiles, tci = rest - tco * a.n_ci_tiles;
    const int t = q >> 3, i = (q >> 1) & 3, j = q & 1;
    const int pw = t2 >> 6, l2 = t2 & 63, wm = pw >> 1, wn = pw & 1;
    const int co = tco * 128 + wm * 64 + i * 16 + (l2 >> 4) * 4;
    const int ci = tci * 64 + wn * 32 + j * 16 + (l2 & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (co + r < a.Cout) a.dw[((size_t)(co + r) * 9 + kh * 3 + t) * a.Cin + ci] += sum[r];
  }
}

__global__ __launch_bounds__(256) void wgrad3_reduce_kernel(WgradArgs a, int G) {
  __shared__ f32x4_t red[4][64];
  const int units = 3 * a.n_co_tiles * a.n_ci_tiles * (SLAB_FLOATS / 256) * G;
  // a capped grid walks the (chunk, slice group) units; every wave of a workgroup makes the same number of rounds
  for (int u0 = blockIdx.x * 4; u0 < units; u0 += gridDim.x * 4) wgrad3_reduce_round(a, G, u0, red);
}

__global__ __launch_bounds__(256) void wgrad3_reduce_group_kernel(WgradGroup g) {
  __shared__ f32x4_t red[4][64];
  const int units = g.ustart[g.n];
  int l = 0;
  for (int u0 = blockIdx.x * 4; u0 < units; u0 += gridDim.x * 4) {   // u0 only grows: l never has to go back
    while (l + 1 < g.n && u0 >= g.ustart[l + 1]) ++l;
    wgrad3_reduce_round(g.it[l], g.G[l], u0 - g.ustart[l], red);
  }
}

void make_magic(unsigned d, unsigned& magic, unsigned& shift) {
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  magic = (unsigned)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  shift = s;
}

}  // namespace

// The launch plan of a layer: kernel variant, tiles and pixel-axis split.
struct WgradPlan {
  bool row3, row9, wide_co, wide_ci;
  int n_co_tiles, n_ci_tiles, per, splitk, rows_per_split, n_blocks;
  size_t ws_bytes;   // workspace for the partial tiles
};

// workgroups of a slab-sum kernel for `units` wave-sized work items (four per workgroup), capped
static int reduce_grid(long units) {
  const long g = (units + 3) / 4;
  return (int)(g < g_dm_wgrad_reduce_cap ? g : g_dm_wgrad_reduce_cap);
}

static int cu_count() {
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  return n_cu;
}

// tiles of a layer (everything of the plan but the pixel-axis split)
static WgradPlan plan_tiles(const dm_conv_desc* d, bool allow9 = true) {
  WgradPlan p;
  const long M = (long)d->B * d->Ho * d->Wo;
  const int Cin = d->C1 + d->C2;
  p.row3 = g_dm_use_wgrad3 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Wo >= 8 &&
           BKM % d->Wo == 0 && M % BKM == 0 && (BKM / d->Wo) * (d->Wo + 2) <= 128;
  // all nine taps per workgroup: a step (64 / Wo output rows) inside one image, its patch within 144 pixels
  // (g_dm_use_wgrad9: 0 off, 1 layers of >= 512 input channels, 3 - the default - also >= 256 channels on images of
  // >= 16x16, 2 wherever the shape allows. Alone the kernel is 7-19 % faster than the three-tap one from 256 input
  // channels up and slower below, where its larger partial tiles cost more than the staging saves; a training step
  // takes the same time within 0.1 % either way.)
  p.row9 = p.row3 && allow9 && g_dm_use_wgrad9 && d->Ho % (BKM / d->Wo) == 0 &&
           (BKM / d->Wo + 2) * (d->Wo + 2) <= 144 && d->C1 % 32 == 0 && d->C2 % 32 == 0 &&
           (g_dm_use_wgrad9 == 2 || Cin >= 512 || (g_dm_use_wgrad9 == 3 && Cin >= 256 && M >= 32768));
  p.wide_co = d->Cout % 128 == 0;
  p.wide_ci = d->C1 % 128 == 0 && d->C2 % 128 == 0;
  const int tco = (p.wide_co || p.row3) ? 128 : 64, tci = p.row9 ? 32 : ((p.wide_ci && !p.row3) ? 128 : 64);
  p.n_co_tiles = dm_cdiv(d->Cout, tco);
  p.n_ci_tiles = Cin / tci;
  p.per = (p.row9 ? 1 : (p.row3 ? 3 : d->KH * d->KW)) * p.n_co_tiles * p.n_ci_tiles;
  return p;
}

// the split once the number of 64-pixel steps per workgroup is chosen
static void plan_split(const dm_conv_desc* d, WgradPlan& p, int splitk) {
  const long M = (long)d->B * d->Ho * d->Wo;
  const int steps_total = dm_cdiv(M, BKM);
  const int tco = (p.wide_co || p.row3) ? 128 : 64, tci = (p.wide_ci && !p.row3) ? 128 : 64;
  p.rows_per_split = dm_cdiv(steps_total, splitk) * BKM;
  p.splitk = dm_cdiv(M, p.rows_per_split);
  p.n_blocks = p.splitk * p.per;
  p.ws_bytes = (size_t)p.n_blocks * (p.row9 ? SLAB9_FLOATS : (p.row3 ? SLAB_FLOATS : tco * tci)) * sizeof(float);
}

static WgradPlan make_plan(const dm_conv_desc* d) {
  WgradPlan p = plan_tiles(d);
  const long M = (long)d->B * d->Ho * d->Wo;
  const int steps_total = dm_cdiv(M, BKM);
  const int n_cu = cu_count();
  // Split the pixel axis to minimise rounds x (steps per workgroup + fixed cost): one workgroup per CU per round,
  // the fixed cost (pipeline fill, partial tile out) counted as OVERHEAD steps.
  const int OVERHEAD = g_dm_wgrad_slab_overhead;
  const int cap = 128;
  const int max_split = steps_total / 8 > 0 ? (steps_total / 8 < cap ? steps_total / 8 : cap) : 1;
  int splitk = 1;
  long best = -1;
  for (int sk = 1; sk <= max_split; ++sk) {
    const long cost = (long)dm_cdiv(p.per * sk, n_cu) * (dm_cdiv(steps_total, sk) + OVERHEAD);
    if (best < 0 || cost < best) { best = cost; splitk = sk; }
  }
  plan_split(d, p, splitk);
  return p;
}

// Plans of n three-tap layers that share one launch: one length T (steps per workgroup) for all of them, chosen to
// minimise rounds x (T + fixed cost) over the summed workgroup count; a layer of S steps is cut into ceil(S / T)
// slices (at most 128, at least 8 steps each, as in the single-layer plan).
static void make_group_plan(const dm_conv_desc* const* ds, int n, WgradPlan* plans) {
  const int n_cu = cu_count(), OVERHEAD = g_dm_wgrad_slab_overhead;
  int steps[WG_MAX], max_steps = 1;
  for (int l = 0; l < n; ++l) {
    plans[l] = plan_tiles(ds[l], false);
    steps[l] = dm_cdiv((long)ds[l]->B * ds[l]->Ho * ds[l]->Wo, BKM);
    if (steps[l] > max_steps) max_steps = steps[l];
  }
  auto split_of = [&](int l, int T) {
    int sk = dm_cdiv(steps[l], T);
    const int most = steps[l] / 8 > 0 ? (steps[l] / 8 < 128 ? steps[l] / 8 : 128) : 1;
    return sk < 1 ? 1 : (sk > most ? most : sk);
  };
  int bestT = max_steps;
  long best = -1;
  for (int T = max_steps; T >= 8; --T) {   // ties go to the longer slices (fewer partial tiles)
    long blocks = 0, longest = 0;
    for (int l = 0; l < n; ++l) {
      const int sk = split_of(l, T);
      blocks += (long)plans[l].per * sk;
      const long len = dm_cdiv(steps[l], sk);
      if (len > longest) longest = len;
    }
    const long cost = dm_cdiv(blocks, n_cu) * (longest + OVERHEAD);
    if (best < 0 || cost < best) { best = cost; bestT = T; }
  }
  for (int l = 0; l < n; ++l) plan_split(ds[l], plans[l], split_of(l, bestT));
}

extern "C" size_t dm_conv2d_wgrad_ws_bytes(const dm_conv_desc* d) {
  return d ? make_plan(d).ws_bytes : 0;
}

template <int TCO, int TCI>
static void launch_single_tap(const WgradArgs& a, const WgradPlan& p, dim3 grid, hipStream_t st) {
  if (g_dm_timing_skip & 4) return;
  hipLaunchKernelGGL((conv_wgrad_kernel<TCO, TCI, 4, 2>), grid, dim3(512), 0, st, a);
  const int G = a.splitk >= 8 ? 4 : (a.splitk >= 4 ? 2 : 1);   // waves sharing a chunk: at least two slices each
  if (g_dm_timing_skip & 2) return;
  hipLaunchKernelGGL((wgrad_reduce_kernel<TCO, TCI>), dim3(reduce_grid((long)p.per * (TCO * TCI / 256) * G)), dim3(256), 0, st, a, G);
}

// argument checks of one layer and the fields of WgradArgs that do not depend on the plan
static int fill_args(const char* who, const dm_conv_desc* d, const void* x1, const void* x2, const void* dy, float* dw,
                     WgradArgs& a) {
  DM_REQUIRE(d && x1 && dy && dw, DM_ERR_ARG, "%s: null argument", who);
  DM_REQUIRE(d->C1 > 0 && d->C1 % 64 == 0 && d->C2 >= 0 && d->C2 % 64 == 0, DM_ERR_SHAPE,
             "%s: input channels (%d, %d) must be multiples of 64", who, d->C1, d->C2);
  DM_REQUIRE((d->C2 == 0) == (x2 == nullptr), DM_ERR_ARG, "%s: x2 and C2 disagree", who);
  DM_REQUIRE(d->Cout > 0 && d->Cout % 8 == 0, DM_ERR_SHAPE, "%s: Cout=%d must be a multiple of 8", who, d->Cout);
  DM_REQUIRE(!d->transposed, DM_ERR_UNSUPPORTED, "%s: transposed descriptor", who);
  DM_REQUIRE(d->stride == 1 || d->stride == 2, DM_ERR_UNSUPPORTED, "%s: stride %d", who, d->stride);
  DM_REQUIRE(!(d->ups && d->stride != 1), DM_ERR_UNSUPPORTED, "%s: fused upsample needs stride 1", who);
  DM_REQUIRE(((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)dy | (uintptr_t)dw) % 16 == 0, DM_ERR_ARG, "%s: pointers must be 16-byte aligned", who);
  const int Hi = d->Hs << d->ups, Wi = d->Ws << d->ups;
  DM_REQUIRE((Hi + 2 * d->pad - d->KH) / d->stride + 1 == d->Ho && (Wi + 2 * d->pad - d->KW) / d->stride + 1 == d->Wo,
             DM_ERR_SHAPE, "%s: in %dx%d k%d s%d p%d does not give out %dx%d", who, Hi, Wi, d->KH, d->stride, d->pad, d->Ho, d->Wo);
  const long npix_src = (long)d->B * d->Hs * d->Ws;
  const long M = (long)d->B * d->Ho * d->Wo;
  DM_REQUIRE(npix_src * d->C1 * 2 < 0x7fffffffL && npix_src * d->C2 * 2 < 0x7fffffffL && M * d->Cout * 2 < 0x7fffffffL,
             DM_ERR_UNSUPPORTED, "%s: tensor larger than 2 GiB", who);
  a.x1 = x1; a.x2 = x2; a.dy = dy; a.dw = dw;
  a.C1 = d->C1; a.C2 = d->C2; a.Cin = d->C1 + d->C2; a.Cout = d->Cout;
  a.B = d->B; a.Hs = d->Hs; a.Ws = d->Ws; a.Hi = Hi; a.Wi = Wi; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.ups = d->ups;
  a.M = (int)M;
  a.x1_bytes = (unsigned)(npix_src * d->C1 * 2);
  a.x2_bytes = (unsigned)(npix_src * d->C2 * 2);
  a.dy_bytes = (unsigned)(M * d->Cout * 2);
  make_magic((unsigned)(d->Ho * d->Wo), a.magic_hw, a.shift_hw);
  make_magic((unsigned)d->Wo, a.magic_w, a.shift_w);
  a.db_step = BKM / (d->Ho * d->Wo);
  a.dh_step = (BKM % (d->Ho * d->Wo)) / d->Wo;
  a.dw_step = (BKM % (d->Ho * d->Wo)) % d->Wo;
  return DM_OK;
}

static void apply_plan(WgradArgs& a, const WgradPlan& p, void* slabs) {
  a.n_co_tiles = p.n_co_tiles; a.n_ci_tiles = p.n_ci_tiles;
  a.rows_per_split = p.rows_per_split; a.splitk = p.splitk; a.n_blocks = p.n_blocks;
  a.slabs = (float*)slabs;
}

static int reduce_waves(int splitk) { return splitk >= 8 ? 4 : (splitk >= 4 ? 2 : 1); }

static void launch_layer(const WgradArgs& a, const WgradPlan& p, hipStream_t st) {
  dim3 grid((a.n_blocks + 7) / 8 * 8);
  if (g_dm_timing_twice & ((p.row3 || p.row9) ? 64 : 4)) {   // timing experiments only: the main kernel once more
    if (p.row9) hipLaunchKernelGGL((conv_wgrad9_kernel<4>), grid, dim3(512), 0, st, a);
    else if (p.row3) hipLaunchKernelGGL((conv_wgrad3_kernel<4>), grid, dim3(512), 0, st, a);
    else if (p.wide_co && p.wide_ci) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 4, 2>), grid, dim3(512), 0, st, a);
  }
  if (p.row9) {
    hipLaunchKernelGGL((conv_wgrad9_kernel<4>), grid, dim3(512), 0, st, a);
    const int G = reduce_waves(a.splitk);
    if (g_dm_timing_skip & 2) return;
    hipLaunchKernelGGL(wgrad9_reduce_kernel, dim3(reduce_grid((long)p.per * (SLAB9_FLOATS / 256) * G)), dim3(256), 0, st, a, G);
  } else if (p.row3) {
    if (g_dm_timing_skip & 64) return;
    hipLaunchKernelGGL((conv_wgrad3_kernel<4>), grid, dim3(512), 0, st, a);
    const int G = reduce_waves(a.splitk);
    if (g_dm_timing_skip & 2) return;
    if (g_dm_timing_twice & 2)
      hipLaunchKernelGGL(wgrad3_reduce_kernel, dim3(reduce_grid((long)p.per * (SLAB_FLOATS / 256) * G)), dim3(256), 0, st, a, G);
    hipLaunchKernelGGL(wgrad3_reduce_kernel, dim3(reduce_grid((long)p.per * (SLAB_FLOATS / 256) * G)), dim3(256), 0, st, a, G);
  } else if (p.wide_co && p.wide_ci) launch_single_tap<128, 128>(a, p, grid, st);
  else if (p.wide_co) launch_single_tap<128, 64>(a, p, grid, st);
  else if (p.wide_ci) launch_single_tap<64, 128>(a, p, grid, st);
  else launch_single_tap<64, 64>(a, p, grid, st);
}

extern "C" int dm_conv2d_wgrad(const dm_conv_desc* d, const void* x1, const void* x2, const void* dy, float* dw, void* ws,
                               size_t ws_bytes, void* stream) {
  WgradArgs a;
  const int rc = fill_args("dm_conv2d_wgrad", d, x1, x2, dy, dw, a);
  if (rc) return rc;
  DM_REQUIRE(ws, DM_ERR_ARG, "dm_conv2d_wgrad: null workspace");
  const WgradPlan p = make_plan(d);
  DM_REQUIRE(ws_bytes >= p.ws_bytes, DM_ERR_ARG, "dm_conv2d_wgrad: workspace of %zu bytes, %zu needed", ws_bytes, p.ws_bytes);
  DM_REQUIRE(((uintptr_t)ws) % 16 == 0, DM_ERR_ARG, "dm_conv2d_wgrad: workspace must be 16-byte aligned");
  apply_plan(a, p, ws);
  launch_layer(a, p, (hipStream_t)stream);
  DM_CHECK_LAUNCH();
  return DM_OK;
}

// ---- several layers per call -------------------------------------------------------------------------------------
// Walks the items in order: runs of up to WG_MAX consecutive three-tap layers share one launch (and one slab-sum
// launch), every other layer is launched as dm_conv2d_wgrad would. With st == nullptr only the workspace is sized.
static int wgrad_group(const dm_wgrad_item* items, int n, char* ws, size_t ws_bytes, size_t* need, bool launch, hipStream_t st) {
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  int i = 0;
  while (i < n) {
    int run = 0;
    while (i + run < n && run < WG_MAX && plan_tiles(&items[i + run].desc, false).row3) ++run;
    if (run <= 1) {   // a layer of another kind, or a three-tap layer with no neighbour to share with
      const dm_wgrad_item& it = items[i];
      const WgradPlan p = make_plan(&it.desc);
      const size_t o = take(p.ws_bytes);
      if (launch) {
        WgradArgs a;
        const int rc = fill_args("dm_conv2d_wgrad_group", &it.desc, it.x1, it.x2, it.dy, it.dw, a);
        if (rc) return rc;
        DM_REQUIRE(off <= ws_bytes, DM_ERR_ARG, "dm_conv2d_wgrad_group: workspace of %zu bytes is too small", ws_bytes);
        apply_plan(a, p, ws + o);
        launch_layer(a, p, st);
      }
      i += 1;
      continue;
    }
    const dm_conv_desc* ds[WG_MAX];
    WgradPlan plans[WG_MAX];
    for (int l = 0; l < run; ++l) ds[l] = &items[i + l].desc;
    make_group_plan(ds, run, plans);
    WgradGroup g;
    g.n = run;
    g.start[0] = 0;
    g.ustart[0] = 0;
    for (int l = 0; l < run; ++l) {
      const size_t o = take(plans[l].ws_bytes);
      g.G[l] = reduce_waves(plans[l].splitk);
      g.start[l + 1] = g.start[l] + plans[l].n_blocks;
      g.ustart[l + 1] = g.ustart[l] + (plans[l].per * (SLAB_FLOATS / 256) * g.G[l] + 3) / 4 * 4;
      if (launch) {
        const dm_wgrad_item& it = items[i + l];
        const int rc = fill_args("dm_conv2d_wgrad_group", &it.desc, it.x1, it.x2, it.dy, it.dw, g.it[l]);
        if (rc) return rc;
        DM_REQUIRE(off <= ws_bytes, DM_ERR_ARG, "dm_conv2d_wgrad_group: workspace of %zu bytes is too small", ws_bytes);
        apply_plan(g.it[l], plans[l], ws + o);
      }
    }
    if (launch) {
      hipLaunchKernelGGL((conv_wgrad3_group_kernel<4>), dim3((g.start[run] + 7) / 8 * 8), dim3(512), 0, st, g);
      hipLaunchKernelGGL(wgrad3_reduce_group_kernel, dim3(reduce_grid(g.ustart[run])), dim3(256), 0, st, g);
    }
    i += run;
  }
  *need = off;
  return DM_OK;
}

extern "C" size_t dm_conv2d_wgrad_group_ws_bytes(const dm_wgrad_item* items, int n) {
  size_t need = 0;
  if (!items || n <= 0 || wgrad_group(items, n, nullptr, 0, &need, false, nullptr) != DM_OK) return 0;
  return need;
}

extern "C" int dm_conv2d_wgrad_group(const dm_wgrad_item* items, int n, void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(items && n > 0 && ws, DM_ERR_ARG, "dm_conv2d_wgrad_group: null argument or no items");
  DM_REQUIRE(((uintptr_t)ws) % 256 == 0, DM_ERR_ARG, "dm_conv2d_wgrad_group: workspace must be 256-byte aligned");
  size_t need = 0;
  const int rc = wgrad_group(items, n, (char*)ws, ws_bytes, &need, true, (hipStream_t)stream);
  if (rc) return rc;
  DM_CHECK_LAUNCH();
  return DM_OK;
}
