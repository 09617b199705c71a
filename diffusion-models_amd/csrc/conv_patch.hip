// 3x3 stride-1 convolution (forward, and data gradient of such a conv) as implicit GEMM on bf16 MFMA with
// LDS-STAGED INPUT PATCHES.
//
// Why a second conv kernel: a CU moves only ~70 GB/s (~29 B/clk) from L2 into LDS, and a 128x128x64 GEMM step
// needs 32 KB of operands for 512 MFMA-cycles, so a plain gathered implicit GEMM tops out near 45 % of the MFMA
// rate. A 3x3 conv re-reads every input pixel for 9 taps; here a tile's input pixels (with their halo) are brought
// into LDS ONCE per 64-channel chunk and all 9 taps multiply shifted views of that patch, and the tile is 256
// pixels tall so each weight byte is used twice as often. Operand traffic per MFMA drops ~3x.
//
// Tile: 256 output pixels (whole image rows, possibly several small images) x BN output channels, 8 waves (4x2),
// each wave 64 x BN/2. Per 64-channel chunk: patch [rows+halo][W+2][64] (zero halo through the buffer range check),
// double buffered; weights [BN][64] per tap through a 3-deep ring. All loads are buffer_load ... lds with counted
// vmcnt waits (loads stay in flight across the per-tap barrier); 16-byte chunks are XOR-swizzled by pixel / row
// index on the source side so fragment reads (ds_read_b128) are conflict free.
//
// Patch geometry. Output pixels of a tile are TR consecutive rows of the "stacked" image (all B*H rows one after the
// other), so they are one contiguous range of the NHWC tensor. In LDS every image that the tile touches gets its own
// zero row above and below, so a tap shift never crosses from one image into the next:
//   patch row of stacked row r = (r - R0) + 1 + 2 * (image(r) - image(R0)),   patch col of column x = x + 1
// and tap (kh, kw) reads patch pixel (row - 1 + kh, col - 1 + kw): a lane-independent offset.
#include "dm_common.h"
#include <type_traits>

namespace {

constexpr int BM = 256;
constexpr int BK = 64;
constexpr int NTHREADS = 512;

struct PatchArgs {
  const void* x1;
  const void* x2;
  const void* w;
  const float* bias;
  const void* res;
  void* y;
  int C1, C2, Cin, Cout;
  int B, H, W, Hs, Ws, ups, transposed;
  int TR, PW;  // rows per tile, patch width (W + 2)
  int M, n_mtiles, n_ntiles;
  unsigned x1_bytes, x2_bytes, w_bytes;
  // optional GroupNorm(+scale/shift)(+SiLU) of the output in the epilogue (gn_y != null): the tile must hold whole
  // samples (BM % gn_hw == 0) and whole groups (BN % gn_cg == 0); y and gn_mean / gn_rstd may then be null (inference:
  // only a backward pass reads the raw output and the statistics)
  const float* gn_gamma;
  const float* gn_beta;
  const float* gn_ss;     // [B][2 Cout] or null
  const float* gn_pre;    // [B][Cout] added to the output before the statistics, or null (not both)
  void* gn_y;
  float* gn_mean;
  float* gn_rstd;
  int gn_hw, gn_G, gn_cg, gn_act;
  float gn_eps;
  // optional 1x1 projection of a second input (the block's residual branch, modeling/blocks.py:74-75) accumulated
  // into the same output tile behind the 3x3 taps (pw != null): p1 | p2 are [B][H][W][PC1 | PC2], pw is [Cout][PCin]
  const void* p1;
  const void* p2;
  const void* pw;
  const float* pbias;
  int PC1, PC2, PCin;
  unsigned p1_bytes, p2_bytes, pw_bytes;
  int direct;   // plain output (no GroupNorm epilogue) straight from the accumulator registers
};

template <int BN, int NP>
struct Lay {
  static constexpr int PIX_CAP = NP * NTHREADS / 8;  // patch pixels per buffer
  static constexpr int A_BYTES = PIX_CAP * BK * 2;
  static constexpr int B_BYTES = BN * BK * 2;
  static constexpr int A0 = 0, A1 = A_BYTES, B0 = 2 * A_BYTES;
  static constexpr int STAGE_TOTAL = 2 * A_BYTES + 3 * B_BYTES;
  static constexpr int EP_PITCH = BN + 4;
  static constexpr int EP_BYTES = BM * EP_PITCH * 4;
  static constexpr int GN_BYTES = 20 * 1024;   // fused GroupNorm: per-wave channel sums, then ka | kb per (sample, channel)
  static constexpr int TOTAL = STAGE_TOTAL > EP_BYTES + GN_BYTES ? STAGE_TOTAL : EP_BYTES + GN_BYTES;
  static constexpr int B_PIECES = BN * 8 / NTHREADS;  // 16-byte pieces of a weight stage per thread
};

// one A-patch piece is issued in tap steps 0..NP-1 of every chunk; loads younger than the weight stage a step waits for
template <int BP, int NP>
__device__ __forceinline__ constexpr int younger_loads(int tap) {
  const int t1 = (tap + 8) % 9, t2 = (tap + 7) % 9;
  return BP + (t1 < NP ? 1 : 0) + (t2 < NP ? 1 : 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Fragment reads as inline asm, issued in one burst per tap step, with waits by hand (the compiler's own schedule
// trickles the reads out between the MFMAs and waits lgkmcnt(0) half a dozen times per step).
__device__ __forceinline__ void lds_read128(f32x4_t& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
}
template <int N, int MI_, int NJ_>
__device__ __forceinline__ void frags_wait(f32x4_t (&a)[MI_], f32x4_t (&b)[NJ_]) {
  static_assert(MI_ == 4 && (NJ_ == 4 || NJ_ == 2), "fragment counts");
  if constexpr (NJ_ == 4)
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1])
                 : "n"(N));
}

// PROJ: the instantiation that also carries the fused 1x1 projection (a.pw != null). Its own instantiation, so that the
// plain kernels keep their register count (what a CU has left beside two of their waves per SIMD is what the other
// stream's HBM-bound kernels run in).
template <int BN, int NP, bool PROJ>
__global__ __launch_bounds__(NTHREADS, 2) void conv_patch_kernel(PatchArgs a) {
  using L = Lay<BN, NP>;
  constexpr int WN = BN / 2, NJ = WN / 16, MI = 4, BP = L::B_PIECES;
  __shared__ __attribute__((aligned(16))) char smem[L::TOTAL];

  int mt, nt;
  dm_xcd_tile(blockIdx.x, a.n_ntiles, mt, nt);
  if (mt >= a.n_mtiles) return;
  const int n0 = nt * BN;
  const int R0 = mt * a.TR;            // first stacked row of the tile
  const int b0 = R0 / a.H, y0 = R0 - b0 * a.H;
  const int nrow0 = min(a.H - y0, a.TR);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // n / d for 0 <= n < 2^20 as a float multiply (exact: (n + 0.5) / d is at least 0.5 / d away from an integer, far
  // more than the rounding error of the product); the index set-up below would otherwise spend thousands of cycles
  // in two dozen integer divisions per thread before the first load is issued
  auto fdiv = [](int n, float inv) { return (int)(((float)n + 0.5f) * inv); };
  const float inv_pw = 1.0f / (float)a.PW, inv_h2 = 1.0f / (float)(a.H + 2), inv_w = 1.0f / (float)a.W,
              inv_h = 1.0f / (float)a.H;

  __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x1), 0, a.x1_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x2 ? a.x2 : a.x1), 0,
                                                                 a.x2 ? a.x2_bytes : 0u, 0x00020000);
  __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);

  // ---- patch staging: this thread's NP pieces = fixed 16-byte slot (tid & 7) of patch pixels (tid >> 3) + 64 * i
  const int spix = tid >> 3;
  // 16-byte slot s of patch pixel (row, column pc) holds channel piece s ^ (pc & 7). Keyed on the COLUMN, not the
  // pixel index: the 16 rows of an MFMA tile are 16 consecutive pixels of one image row or, on 8-wide images, 8 + 8
  // pixels of two image rows with equal columns, and either way the 8 lanes that one LDS cycle serves per 32-channel
  // half see 8 different columns mod 8 (a pixel-index key collides on 8-wide images: the row pitch is 10).
  int akey[NP];
  int gsrc[NP];                                // source pixel index, -1: zero
  unsigned own = 0;                            // bit i: piece i is a pixel of the tile itself (not a halo pixel)
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pp = spix + 64 * i;
    const int pr = fdiv(pp, inv_pw), pc = pp - pr * a.PW;
    akey[i] = (tid & 7) ^ (pc & 7);
    int k, local, ystart;
    if (pr < nrow0 + 2) { k = 0; local = pr; ystart = y0; }
    else {
      const int q = pr - (nrow0 + 2);
      k = 1 + fdiv(q, inv_h2);
      local = q - (k - 1) * (a.H + 2);
      ystart = 0;
    }
    const int yin = ystart - 1 + local, xin = pc - 1, b = b0 + k;
    const bool ok = xin >= 0 && xin < a.W && yin >= 0 && yin < a.H && b < a.B;
    gsrc[i] = ok ? ((b * a.Hs + (yin >> a.ups)) * a.Ws + (xin >> a.ups)) : -1;
    if (PROJ && ok && local >= 1 && local <= (k == 0 ? nrow0 : a.H)) own |= 1u << i;
  }
  // Row r of a staged weight tile holds output channel n0 + chan_of_row(r): inside a wave's WN channels, row m of
  // MFMA tile j is channel (j / 2) * 32 + (m / 4) * 8 + (j % 2) * 4 + m % 4. The product is formed transposed
  // (weights as the MFMA's row operand), so accumulator register r of tile j in lane group fq is row 4 fq + r of
  // that tile: with this deal a lane holds 8 CONSECUTIVE channels (tiles 2g, 2g + 1) of a pixel, and the output can
  // leave the registers as 16-byte stores - 64 contiguous bytes per pixel and store instruction - without LDS.
  auto chan_of_row = [](int r) {
    const int base = r / WN * WN, lr = r - base, j = lr >> 4, m = lr & 15;
    return base + (j >> 1) * 32 + (m >> 2) * 8 + (j & 1) * 4 + (m & 3);
  };
  unsigned wrow[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int n = n0 + chan_of_row((tid >> 3) + 64 * i);
    const int chunkB = (tid & 7) ^ ((tid >> 3) & 7);
    wrow[i] = (n < a.Cout) ? (unsigned)(n * 9 * a.Cin * 2 + chunkB * 16) : DM_OOB;
  }

  auto issue_A_piece = [&](int i, int cc, int buf) {   // piece i of the patch for channel chunk cc
    unsigned off = DM_OOB;
    const bool second = cc >= a.C1;
    if (cc < a.Cin && gsrc[i] >= 0) {
      const int csrc = second ? a.C2 : a.C1;
      off = (unsigned)(gsrc[i] * csrc * 2 + (second ? cc - a.C1 : cc) * 2 + akey[i] * 16);
    }
    char* dst = smem + (buf ? L::A1 : L::A0) + (wave * 64 + NTHREADS * i) * 16;
    if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, DM_LDS(dst), 16, off, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, DM_LDS(dst), 16, off, 0, 0, 0);
  };
  auto issue_B = [&](int tap, int cc, int stage) {     // weights of (tap, chunk cc); cc >= Cin: nothing (zeros)
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      unsigned off = (wrow[i] == DM_OOB || cc >= a.Cin) ? DM_OOB : wrow[i] + (unsigned)((tap * a.Cin + cc) * 2);
      char* dst = smem + L::B0 + stage * L::B_BYTES + (wave * 64 + NTHREADS * i) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, DM_LDS(dst), 16, off, 0, 0, 0);
    }
  };

  // ---- fragment geometry
  const int frow = lane & 15, fq = lane >> 4;
  int pA[MI];       // patch pixel of this lane's row of M-tile i at tap offset 0
  int xA[MI];       // its column in the tile (patch column at tap offset 0)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int lp = wm * 64 + i * 16 + frow;
    const int ry = fdiv(lp, inv_w), x = lp - ry * a.W;
    const int r = R0 + ry;
    if (ry < a.TR && r < a.B * a.H) pA[i] = (ry + 2 * (fdiv(r, inv_h) - b0)) * a.PW + x;
    else pA[i] = 0;  // rows outside the tile compute garbage that is never stored
    xA[i] = x;
  }
  int b_off[NJ][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = wn * WN + j * 16 + frow;
      b_off[j][kk] = row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4);
    }

  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: patch of chunk 0, weights of steps 0 and 1
#pragma unroll
  for (int i = 0; i < NP; ++i) issue_A_piece(i, 0, 0);
  issue_B(0, 0, 0);
  issue_B(1, 0, 1);

  const unsigned lds_base = (unsigned)(size_t)DM_LDS(smem);
  f32x4_t af[2][MI], bfr[2][NJ];  // the fragments of one tap step (both 32-channel halves)
  auto multiply = [&]() {
    frags_wait<MI + NJ>(af[0], bfr[0]);   // the second half's reads may still be in flight
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk == 1) frags_wait<0>(af[1], bfr[1]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bfr[kk][j]),
                                                              __builtin_bit_cast(bf16x8_t, af[kk][i]), acc[i][j], 0, 0, 0);
    }
  };
  constexpr bool SHIFT = BN == 128;
  const bool lead = !SHIFT || __builtin_amdgcn_readfirstlane(wave) < 4;

  const int nchunks = a.Cin / BK;
  for (int c = 0; c < nchunks; ++c) {
    const int cc = c * BK;
    const unsigned abase = lds_base + ((c & 1) ? L::A1 : L::A0);
    // keep the 72 per-tap fragment addresses from being hoisted out of this loop into registers
    int pw = a.PW;
    asm volatile("" : "+s"(pw));
    auto step = [&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      // weights of this step (and everything older, i.e. this chunk's patch) have landed for this wave ...
      wait_vmcnt<younger_loads<BP, NP>(tap)>();
      // ... and for every wave; every wave has also finished reading the previous step's weight stage (the second
      // wave group issued those reads last thing in that step: drain them before anybody refills the stage)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // 128-wide tiles: the two waves of a SIMD (wave groups 0-3 / 4-7) run half a step out of phase - the first
      // group reads this step's fragments and multiplies them, the second multiplies what it read in the previous
      // step, then reads - so one wave's LDS latency and address arithmetic sit under the other's MFMAs (measured
      // +3..6 % inside a training step). 64-wide tiles have half the MFMAs per step to hide anything under and are
      // faster with the plain read-then-multiply order.
      if constexpr (SHIFT) {
        if (!lead && !(tap == 0 && c == 0)) multiply();
      }
      // weights two steps ahead, and one piece of the next chunk's patch
      {
        const int t2 = (tap + 2) % 9;
        issue_B(t2, tap + 2 >= 9 ? cc + BK : cc, (tap + 2) % 3);
        if (tap < NP) issue_A_piece(tap, cc + BK, (c + 1) & 1);
      }
      const int kh = tap / 3, kw = tap - kh * 3;
      const int toff = a.transposed ? ((2 - kh) * pw + (2 - kw)) : (kh * pw + kw);
      const int coff = a.transposed ? 2 - kw : kw;   // column part of the tap offset
      if constexpr (SHIFT) {
        const unsigned bbase = lds_base + L::B0 + (tap % 3) * L::B_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int p = pA[i] + toff;
            lds_read128(af[kk][i], abase + (p << 7) + ((((kk * 4 + fq) ^ (xA[i] + coff)) & 7) << 4));
          }
#pragma unroll
          for (int j = 0; j < NJ; ++j) lds_read128(bfr[kk][j], bbase + b_off[j][kk]);
        }
        if (lead) multiply();
      } else {
        const char* abuf = smem + ((c & 1) ? L::A1 : L::A0);
        const char* bbuf = smem + L::B0 + (tap % 3) * L::B_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8_t a2[MI], b2[NJ];
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int p = pA[i] + toff;
            a2[i] = *(const bf16x8_t*)(abuf + (p << 7) + ((((kk * 4 + fq) ^ (xA[i] + coff)) & 7) << 4));
          }
#pragma unroll
          for (int j = 0; j < NJ; ++j) b2[j] = *(const bf16x8_t*)(bbuf + b_off[j][kk]);
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2[j], a2[i], acc[i][j], 0, 0, 0);
        }
      }
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});
    step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});
  }

  if (SHIFT && !lead) multiply();

  // ---- the residual branch's 1x1 projection, accumulated into the same tile: per 64-channel chunk of the second
  // input the tile's own pixels go into a patch buffer (halo pieces load zeros without touching memory) and the
  // projection's weights into a ring stage; the centre tap's fragment addresses read them. One chunk ahead, both
  // wave groups in phase: the chunk is bound by its loads (48 KB for 32 MFMAs per wave), not by the matrix core.
  if constexpr (PROJ) {
    __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.p1), 0, a.p1_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rp2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.p2 ? a.p2 : a.p1), 0,
                                                                   a.p2 ? a.p2_bytes : 0u, 0x00020000);
    __amdgpu_buffer_rsrc_t rpw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pw), 0, a.pw_bytes, 0x00020000);
    unsigned prow[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int n = n0 + chan_of_row((tid >> 3) + 64 * i);
      const int chunkB = (tid & 7) ^ ((tid >> 3) & 7);
      prow[i] = (n < a.Cout) ? (unsigned)(n * a.PCin * 2 + chunkB * 16) : DM_OOB;
    }
    auto issue_proj = [&](int cc, int buf) {
      const bool second = cc >= a.PC1;
      const int csrc = second ? a.PC2 : a.PC1, coff = second ? cc - a.PC1 : cc;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const unsigned off = ((own >> i) & 1u) ? (unsigned)(gsrc[i] * csrc * 2 + coff * 2 + akey[i] * 16) : DM_OOB;
        char* dst = smem + (buf ? L::A1 : L::A0) + (wave * 64 + NTHREADS * i) * 16;
        if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rp2, DM_LDS(dst), 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rp1, DM_LDS(dst), 16, off, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const unsigned off = prow[i] == DM_OOB ? DM_OOB : prow[i] + (unsigned)(cc * 2);
        char* dst = smem + L::B0 + buf * L::B_BYTES + (wave * 64 + NTHREADS * i) * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rpw, DM_LDS(dst), 16, off, 0, 0, 0);
      }
    };
    // the main loop's trailing loads have landed and every wave has read its last fragments
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int npc = a.PCin / BK;
    issue_proj(0, 0);
    for (int c = 0; c < npc; ++c) {
      if (c + 1 < npc) {
        issue_proj((c + 1) * BK, (c + 1) & 1);
        wait_vmcnt<NP + BP>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned abase = lds_base + ((c & 1) ? L::A1 : L::A0);
      const unsigned bbase = lds_base + L::B0 + (c & 1) * L::B_BYTES;
      const int toff = a.PW + 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          lds_read128(af[kk][i], abase + ((pA[i] + toff) << 7) + ((((kk * 4 + fq) ^ (xA[i] + 1)) & 7) << 4));
#pragma unroll
        for (int j = 0; j < NJ; ++j) lds_read128(bfr[kk][j], bbase + b_off[j][kk]);
      }
      multiply();
      // every wave is done with this chunk's buffers before the next iteration refills them
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }

  // ---- epilogue (the trailing dummy loads must have landed before the workgroup ends or LDS is reused)
  wait_vmcnt<0>();
  bf16_t* y = (bf16_t*)a.y;
  const bf16_t* res = (const bf16_t*)a.res;
  const int m0 = R0 * a.W;              // the tile's output pixels are contiguous in memory
  const int rows_valid = min(a.TR * a.W, a.M - m0);
  if (a.gn_y == nullptr && a.direct) {
    // straight from the registers: lane (frow, fq) holds, of pixel wm * 64 + i * 16 + frow, the 8 channels
    // n0 + wn * WN + g * 32 + fq * 8 .. of every g (accumulator tiles 2g and 2g + 1)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * 64 + i * 16 + frow;
      if (row >= rows_valid) continue;
#pragma unroll
      for (int g = 0; g < NJ / 2; ++g) {
        const int n = n0 + wn * WN + g * 32 + fq * 8;
        if (n >= a.Cout) continue;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * g][r]; v[4 + r] = acc[i][2 * g + 1][r]; }
        if (a.bias) {
          const f32x4_t c0 = *(const f32x4_t*)(a.bias + n), c1 = *(const f32x4_t*)(a.bias + n + 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] += c0[r]; v[4 + r] += c1[r]; }
        }
        if (PROJ && a.pbias) {
          const f32x4_t c0 = *(const f32x4_t*)(a.pbias + n), c1 = *(const f32x4_t*)(a.pbias + n + 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] += c0[r]; v[4 + r] += c1[r]; }
        }
        if (res) {
          const bf16x8_t rv = *(const bf16x8_t*)(res + (size_t)(m0 + row) * a.Cout + n);
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
        }
        *(bf16x8_t*)(y + (size_t)(m0 + row) * a.Cout + n) = dm_pack8(v);
      }
    }
    return;
  }

  // ---- with a GroupNorm epilogue the tile goes through LDS as [pixel][channel] floats first
  __syncthreads();
  float* ep = (float*)smem;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ep[(wm * 64 + i * 16 + frow) * L::EP_PITCH + wn * WN + (j >> 1) * 32 + fq * 8 + (j & 1) * 4 + r] = acc[i][j][r];
  __syncthreads();
  constexpr int CPR = BN / 8;
  // 8 channels of one output row: accumulator + bias + residual, rounded to bf16
  auto finish_row = [&](int row, int cch) -> bf16x8_t {
    const float* src = ep + row * L::EP_PITCH + cch * 8;
    const int n = n0 + cch * 8;
    const f32x4_t v0 = *(const f32x4_t*)src, v1 = *(const f32x4_t*)(src + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    if (a.bias) {
      const f32x4_t c0 = *(const f32x4_t*)(a.bias + n), c1 = *(const f32x4_t*)(a.bias + n + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += c0[j]; v[4 + j] += c1[j]; }
    }
    if (PROJ && a.pbias) {
      const f32x4_t c0 = *(const f32x4_t*)(a.pbias + n), c1 = *(const f32x4_t*)(a.pbias + n + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += c0[j]; v[4 + j] += c1[j]; }
    }
    if (res && row < rows_valid) {
      const bf16x8_t rv = *(const bf16x8_t*)(res + (size_t)(m0 + row) * a.Cout + n);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
    }
    return dm_pack8(v);
  };
  if (a.gn_y == nullptr) {   // (patch_direct=0: the plain output through LDS as well, for comparison)
#pragma unroll 2
    for (int it = 0; it < (BM * CPR) / NTHREADS; ++it) {
      const int item = tid + NTHREADS * it;
      const int row = item / CPR, cch = item - row * CPR;
      if (row < rows_valid && n0 + cch * 8 < a.Cout)
        *(bf16x8_t*)(y + (size_t)(m0 + row) * a.Cout + n0 + cch * 8) = finish_row(row, cch);
    }
    return;
  }

  // ---- with the GroupNorm (+ per-sample scale/shift or pre-add) (+ SiLU) of the output. The tile holds S = BM / HW
  // whole samples and BN / cg whole groups (the host has checked); every thread keeps its share of the rounded output
  // in registers: y2 = act(ka * y + kb), ka = (1 + scale) gamma rstd, kb = ka (pre - mean) + (1 + scale) beta + shift.
  // All sums run in a fixed order that does not depend on where in a tile (or a batch) a sample sits.
  // Order of the sums over a sample's rows, the same for both tile widths: rows fall into 64 classes (row mod 64), a
  // class adds its rows in ascending order, classes 8w .. 8w+7 (held by wave w) add up as a butterfly over the class
  // bits, the eight waves' sums in ascending order. (64-wide tiles: a thread has one class; 128-wide: two, 2 rg and
  // 2 rg + 1, on alternating iterations.)
  constexpr int RPI = NTHREADS / CPR;   // rows per iteration of the whole workgroup
  constexpr int CLS = 64 / RPI;         // classes per thread
  constexpr int NIT = BM / RPI;
  constexpr int NW = NTHREADS / 64;
  const int HW = a.gn_hw, S = BM / HW, ips = HW / RPI;   // iterations per sample (HW is a multiple of 64)
  float* part = (float*)(smem + L::EP_BYTES);            // [S][NW][2][BN] per-wave channel sums
  float* kab = part + S * NW * 2 * BN;                   // [S][2][BN] ka | kb
  const int cch = tid % CPR, rg = tid / CPR;
  auto row_of = [&](int it) { return CLS * rg + it % CLS + 64 * (it / CLS); };
  bf16x8_t keep[NIT];
  float s1[CLS][8], s2[CLS][8];
#pragma unroll
  for (int k = 0; k < CLS; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[k][j] = s2[k][j] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = row_of(it);
    keep[it] = finish_row(row, cch);
    if (y && row < rows_valid) *(bf16x8_t*)(y + (size_t)(m0 + row) * a.Cout + n0 + cch * 8) = keep[it];
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // statistics of what a separate kernel would read back: the rounded values
      const float f = (float)keep[it][j];
      s1[it % CLS][j] += f;
      s2[it % CLS][j] += f * f;
    }
    if ((it + 1) % ips == 0) {      // the sample's last rows: lanes that share the channels add up, one of them files
      const int sidx = it / ips;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t1 = s1[0][j], t2 = s2[0][j];
        if (CLS == 2) { t1 += s1[CLS - 1][j]; t2 += s2[CLS - 1][j]; }
#pragma unroll
        for (int m = CPR; m < 64; m <<= 1) {
          t1 += __shfl_xor(t1, m);
          t2 += __shfl_xor(t2, m);
        }
        s1[0][j] = t1;
        s2[0][j] = t2;
      }
      if (lane < CPR) {
        float* dst = part + ((sidx * NW + wave) * 2) * BN + cch * 8;
        *(f32x4_t*)dst = f32x4_t{s1[0][0], s1[0][1], s1[0][2], s1[0][3]};
        *(f32x4_t*)(dst + 4) = f32x4_t{s1[0][4], s1[0][5], s1[0][6], s1[0][7]};
        *(f32x4_t*)(dst + BN) = f32x4_t{s2[0][0], s2[0][1], s2[0][2], s2[0][3]};
        *(f32x4_t*)(dst + BN + 4) = f32x4_t{s2[0][4], s2[0][5], s2[0][6], s2[0][7]};
      }
#pragma unroll
      for (int k = 0; k < CLS; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) s1[k][j] = s2[k][j] = 0.f;
    }
  }
  __syncthreads();
  const int b_first = m0 / HW;
  if (tid < S * BN) {   // one thread per (sample, channel); a group's cg channels sit in adjacent lanes of one wave
    const int sidx = tid / BN, c = tid - sidx * BN, cj = n0 + c, b = b_first + sidx;
    const bool live = sidx * HW < rows_valid;
    float u1 = 0.f, u2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      u1 += part[((sidx * NW + w) * 2) * BN + c];
      u2 += part[((sidx * NW + w) * 2 + 1) * BN + c];
    }
    const float pv = (a.gn_pre && live) ? a.gn_pre[(size_t)b * a.Cout + cj] : 0.f;
    u2 += 2.f * pv * u1 + (float)HW * pv * pv;   // statistics of y + pre from those of y
    u1 += (float)HW * pv;
    for (int m = 1; m < a.gn_cg; m <<= 1) {      // (every lane of a group ends with the same bits: a + b == b + a)
      u1 += __shfl_xor(u1, m);
      u2 += __shfl_xor(u2, m);
    }
    const float inv_n = 1.0f / (float)(a.gn_cg * HW);
    const float mu = u1 * inv_n;
    const float var = fmaxf(u2 * inv_n - mu * mu, 0.f);
    const float r = rsqrtf(var + a.gn_eps);
    float ka = 0.f, kb = 0.f;
    if (live) {
      float sc = 1.f, sh = 0.f;
      if (a.gn_ss) { sc = 1.f + a.gn_ss[(size_t)b * 2 * a.Cout + cj]; sh = a.gn_ss[(size_t)b * 2 * a.Cout + a.Cout + cj]; }
      ka = sc * a.gn_gamma[cj] * r;
      kb = ka * (pv - mu) + sc * a.gn_beta[cj] + sh;
      if (c % a.gn_cg == 0 && a.gn_mean) {
        a.gn_mean[(size_t)b * a.gn_G + cj / a.gn_cg] = mu;
        a.gn_rstd[(size_t)b * a.gn_G + cj / a.gn_cg] = r;
      }
    }
    kab[(sidx * 2) * BN + c] = ka;
    kab[(sidx * 2 + 1) * BN + c] = kb;
  }
  __syncthreads();
  bf16_t* y2 = (bf16_t*)a.gn_y;
  float ka[8], kb[8];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = row_of(it);
    if (it % ips == 0) {
      const float* k = kab + ((it / ips) * 2) * BN + cch * 8;
      const f32x4_t a0 = *(const f32x4_t*)k, a1 = *(const f32x4_t*)(k + 4);
      const f32x4_t b0_ = *(const f32x4_t*)(k + BN), b1_ = *(const f32x4_t*)(k + BN + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { ka[j] = a0[j]; ka[4 + j] = a1[j]; kb[j] = b0_[j]; kb[4 + j] = b1_[j]; }
    }
    if (row < rows_valid) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = ka[j] * (float)keep[it][j] + kb[j];
        f[j] = a.gn_act ? dm_silu(u) : u;
      }
      *(bf16x8_t*)(y2 + (size_t)(m0 + row) * a.Cout + n0 + cch * 8) = dm_pack8(f);
    }
  }
}

// most images a run of tr stacked rows of H-row images can touch, given that runs start at multiples of tr
int images_touched(int tr, int H) {
  if (tr <= H && H % tr == 0) return 1;
  if (tr > H && tr % H == 0) return tr / H;
  return (tr + H - 2) / H + 1;
}

// rows per tile such that the tile has <= 256 pixels and its patch (own zero rows per image) fits `cap` pixels
int pick_rows(int H, int W, int cap) {
  for (int tr = BM / W; tr > 0; --tr)
    if ((tr + 2 * images_touched(tr, H)) * (W + 2) <= cap) return tr;
  return 0;
}

}  // namespace

// Returns DM_OK if handled, DM_ERR_UNSUPPORTED (without setting an error) when the shape is not covered and the
// caller should use the gathered implicit-GEMM kernel.
// Tile plan of a layer: (64-wide tiles?, rows per tile); rows = 0: the layer does not take this kernel.
static void patch_plan(const dm_conv_desc* d, bool& bn64, int& TR, bool force64 = false) {
  TR = 0;
  bn64 = false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return;
  const int H = d->Ho, W = d->Wo;
  if (W > BM || d->Cout % 8) return;
  if ((long)d->B * H >= (1 << 20) - BM) return;   // the kernel's float-multiply divisions are exact below 2^20
  const int Cin = d->C1 + d->C2;
  if (Cin % BK || d->C1 % BK) return;
  // narrow outputs on small images: 64-column tiles keep every CU busy
  // (measured with cold caches: with at most 32 row tiles, 64-column tiles win up to 1024 output channels)
  const long M = (long)d->B * H * W;
  const long mt = M / BM, blocks128 = mt * dm_cdiv(d->Cout, 128);
  bn64 = force64 || d->Cout <= 64 || blocks128 < 200 || (mt <= 32 && blocks128 <= 256);
  const int cap = bn64 ? Lay<64, 7>::PIX_CAP : Lay<128, 6>::PIX_CAP;
  const int tr = pick_rows(H, W, cap);
  if (tr <= 0 || tr * W < BM / 2) return;         // too ragged to be worth it
  TR = tr;
}

// Does a tile plan let the GroupNorm that follows the convolution run in its epilogue? The tile must be exactly BM
// pixels of whole samples and BN channels of whole groups, and the workgroup must sit inside one sample at a time.
static bool gn_plan_ok(const dm_conv_desc* d, const dm_gn_desc* g, bool bn64, int TR) {
  if (!TR) return false;
  const int HW = d->Ho * d->Wo, BN = bn64 ? 64 : 128;
  if (g->C1 != d->Cout || g->C2 != 0 || g->HW != HW || g->B != d->B || (g->has_pre && g->has_ss) || g->G <= 0 ||
      d->Cout % g->G)
    return false;
  const int cg = d->Cout / g->G;
  if (TR * d->Wo != BM || BM % HW || d->Cout % BN || BN % cg) return false;
  if (cg > 64 || (cg & (cg - 1))) return false;   // a group's channels add up across 2^k adjacent lanes
  const int S = BM / HW;
  if (HW % 64 || S * BN > NTHREADS) return false;
  const int gn_bytes = (S * (NTHREADS / 64) * 2 * BN + S * 2 * BN) * 4;
  return gn_bytes <= (bn64 ? Lay<64, 7>::GN_BYTES : Lay<128, 6>::GN_BYTES);
}

// The plan for convolution + GroupNorm: the layer's own plan if it qualifies, else 64-wide tiles if those do (whatever
// qualifies with 128-wide tiles also does with 64-wide ones, so whether a layer qualifies does not depend on the
// batch size, and the statistics come out of the same order of sums either way).
static bool gn_plan(const dm_conv_desc* d, const dm_gn_desc* g, bool& bn64, int& TR) {
  patch_plan(d, bn64, TR);
  if (gn_plan_ok(d, g, bn64, TR)) return true;
  if (bn64) return false;
  patch_plan(d, bn64, TR, true);
  return gn_plan_ok(d, g, bn64, TR);
}

int dm_conv_patch_gn_ok(const dm_conv_desc* d, const dm_gn_desc* g) {
  bool bn64;
  int TR;
  return g && gn_plan(d, g, bn64, TR);
}

// Can the 1x1 projection `p` of a second input ride in this layer's patch kernel? (a regular 3x3 convolution that
// takes the patch kernel, no fused upsampling; projection channels in whole 64-channel chunks per source)
int dm_conv_patch_proj_ok(const dm_conv_desc* d, const dm_conv_proj* p) {
  bool bn64;
  int TR;
  if (!d || !p || d->transposed || d->ups) return 0;
  patch_plan(d, bn64, TR);
  if (!TR) return 0;
  if (p->C1 <= 0 || p->C1 % BK || p->C2 < 0 || p->C2 % BK) return 0;
  const long npix = (long)d->B * d->Ho * d->Wo;
  return npix * p->C1 * 2 < 0x7fffffffL && npix * p->C2 * 2 < 0x7fffffffL && (long)d->Cout * (p->C1 + p->C2) * 2 < 0x7fffffffL;
}

extern int g_dm_patch_direct;

int dm_conv_patch_try(const dm_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                      const void* res, void* y, hipStream_t st, const dm_gn_desc* g, const float* gamma,
                      const float* beta, const float* ss, const float* pre, void* y_norm, float* mean,
                      float* rstd, const dm_conv_proj* proj) {
  bool bn64;
  int TR;
  if (g) {
    if (!gn_plan(d, g, bn64, TR)) return DM_ERR_UNSUPPORTED;
  } else {
    patch_plan(d, bn64, TR);
  }
  if (!TR) return DM_ERR_UNSUPPORTED;
  if (proj && !dm_conv_patch_proj_ok(d, proj)) return DM_ERR_UNSUPPORTED;
  const int H = d->Ho, W = d->Wo, Cin = d->C1 + d->C2;
  const long M = (long)d->B * H * W;

  PatchArgs a;
  a.x1 = x1; a.x2 = x2; a.w = w; a.bias = bias; a.res = res; a.y = y;
  a.C1 = d->C1; a.C2 = d->C2; a.Cin = Cin; a.Cout = d->Cout;
  a.B = d->B; a.H = H; a.W = W; a.Hs = d->Hs; a.Ws = d->Ws; a.ups = d->ups; a.transposed = d->transposed;
  a.TR = TR; a.PW = W + 2;
  a.M = (int)M;
  const long npix_src = (long)d->B * d->Hs * d->Ws;
  a.x1_bytes = (unsigned)(npix_src * d->C1 * 2);
  a.x2_bytes = (unsigned)(npix_src * d->C2 * 2);
  a.w_bytes = (unsigned)((long)d->Cout * 9 * Cin * 2);
  a.p1 = a.p2 = a.pw = nullptr;
  a.pbias = nullptr;
  a.PC1 = a.PC2 = a.PCin = 0;
  a.p1_bytes = a.p2_bytes = a.pw_bytes = 0;
  if (proj) {
    a.p1 = proj->x1; a.p2 = proj->x2; a.pw = proj->w; a.pbias = proj->bias;
    a.PC1 = proj->C1; a.PC2 = proj->C2; a.PCin = proj->C1 + proj->C2;
    a.p1_bytes = (unsigned)(M * proj->C1 * 2);
    a.p2_bytes = (unsigned)(M * proj->C2 * 2);
    a.pw_bytes = (unsigned)((long)d->Cout * a.PCin * 2);
  }
  a.direct = g_dm_patch_direct;
  a.gn_y = nullptr;
  a.gn_gamma = a.gn_beta = a.gn_ss = a.gn_pre = nullptr;
  a.gn_mean = a.gn_rstd = nullptr;
  a.gn_hw = a.gn_G = a.gn_cg = a.gn_act = 0;
  a.gn_eps = 0.f;
  if (g) {
    a.gn_gamma = gamma; a.gn_beta = beta; a.gn_ss = ss; a.gn_pre = pre; a.gn_y = y_norm; a.gn_mean = mean; a.gn_rstd = rstd;
    a.gn_hw = H * W; a.gn_G = g->G; a.gn_cg = d->Cout / g->G; a.gn_act = g->act; a.gn_eps = g->eps;
  }
  a.n_mtiles = dm_cdiv((long)d->B * H, TR);
  const int mt_pad = (a.n_mtiles + 7) / 8 * 8;
  if (bn64) {
    a.n_ntiles = dm_cdiv(d->Cout, 64);
    if (proj) hipLaunchKernelGGL((conv_patch_kernel<64, 7, true>), dim3(mt_pad * a.n_ntiles), dim3(NTHREADS), 0, st, a);
    else hipLaunchKernelGGL((conv_patch_kernel<64, 7, false>), dim3(mt_pad * a.n_ntiles), dim3(NTHREADS), 0, st, a);
  } else {
    a.n_ntiles = dm_cdiv(d->Cout, 128);
    if (proj) hipLaunchKernelGGL((conv_patch_kernel<128, 6, true>), dim3(mt_pad * a.n_ntiles), dim3(NTHREADS), 0, st, a);
    else hipLaunchKernelGGL((conv_patch_kernel<128, 6, false>), dim3(mt_pad * a.n_ntiles), dim3(NTHREADS), 0, st, a);
  }
  return DM_OK;
}
