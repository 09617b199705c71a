/* C ABI of libdm_hip.so: hand-written gfx950 (MI355X) kernels for the UNet denoiser training step and
 * sampling loop of Thxios/diffusion-models.
 *
 * The reference has no FFI of its own: every operator below replaces a stock PyTorch call site inside the
 * reference's Python modules; each declaration cites the call site (paths relative to the reference tree).
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller, 16-byte aligned unless noted; nothing is allocated,
 *     freed or synchronised inside; every kernel is launched on `stream` (a hipStream_t) and returns at once
 *   - activations are "pixel-major" (NHWC): a[m][c], m = (b*H + h)*W + w, stored as bf16
 *   - parameters, statistics, embeddings, losses and gradients of parameters are fp32
 *   - return value: 0 (DM_OK) or a dm_status code; dm_last_error() returns the message of the calling thread's
 *     last failure
 */
#ifndef DM_HIP_H
#define DM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dm_status {
  DM_OK = 0,
  DM_ERR_ARG = 1,         /* null / misaligned / inconsistent argument */
  DM_ERR_SHAPE = 2,       /* shape the kernels do not cover (e.g. channels not a multiple of 64) */
  DM_ERR_LAUNCH = 3,      /* the HIP runtime refused the launch */
  DM_ERR_UNSUPPORTED = 4
} dm_status;

const char* dm_last_error(void);

/* ---------------------------------------------------------------------------------------------------------
 * Ordered sums. Parameter gradients that are sums over samples, pixels or workgroups (bias gradients, GroupNorm
 * affine gradients, edge-conv weights) are never accumulated with float atomics: the producing kernel leaves partial
 * vectors in scratch and an ordered sum adds them up in a fixed order, so a training step is bitwise reproducible.
 *   s[e] = sum_q part[q * stride + e], q = 0 .. nparts-1, e < n
 *   e in [0, s1): out0[e] += s (and dual0[e] += s when given); [s1, s2): out1[e - s1] += s; [s2, n): out2[e - s2] += s
 * (a null output skips its segment). The operators below that produce such sums take a `dm_sum_item* deferred`:
 * null = the ordered sum is launched at once on the same stream; non-null = the item is filled in instead and the
 * caller runs any number of them later in ONE launch with dm_ordered_sum_group, on a stream ordered after the
 * producers (the scratch must stay untouched until then). `items` is a HOST array. Items of one call whose outputs
 * overlap (a parameter used twice in one forward pass files two sums for one gradient) are added one after the other,
 * in item order, by separate launches; immediate sums (deferred == null) into one output must be issued on ONE stream. */
typedef struct dm_sum_item {
  const float* part;
  long stride;
  int nparts, n, s1, s2;
  float* out0;
  float* out1;
  float* out2;
  float* dual0;
} dm_sum_item;
int dm_ordered_sum_group(const dm_sum_item* items, int n_items, void* stream);
int dm_abi_version(void);
/* Run-time switches (tuning and A/B measurements; results are the same to the documented tolerances either way):
 *   "conv_patch" (1)      3x3 stride-1 convolutions use the LDS-staged-patch kernel
 *   "conv_gn_fuse" (1)    dm_conv2d_gn_fusable may say yes (GroupNorm in the producing convolution's epilogue)
 *   "conv_proj_fuse" (1)  dm_conv2d_proj_fusable may say yes (a 1x1 projection of a second input inside the 3x3 launch)
 *   "conv1x1_stream" (1)  1x1 stride-1 convolutions with at most 256 input channels over >= 32768 pixels (channels in
 *                         multiples of 128) take the streaming kernel; 2: wherever the channel counts allow; 0: never
 *   "patch_direct" (1)    plain outputs of the patch kernel leave the accumulator registers directly (0: through LDS,
 *                         as the outputs with a GroupNorm epilogue always do); bit-identical either way
 *   "gn_min_ppt" (4)      single-pass GroupNorm kernels: least pixels per thread (1, 2, 4, 8) on images small enough to
 *                         leave 8 pixel lanes - smaller workgroups with more bytes in flight per wave (8x8 images)
 *   "dgrad_subpixel" (1)  stride-2 data gradients run by output-pixel parity class
 *   "wgrad_row3" (1)      3x3 weight gradients share one staged input row among the 3 taps of a kernel row
 *   "wgrad_row9" (3)      ... or all nine taps per workgroup: 0 never, 1 from 512 input channels, 3 also from 256 input
 *                         channels on images of >= 16x16 (7-19 % faster alone there; the training step is the same
 *                         within 0.1 %), 2 wherever the shape allows
 *   "wgrad_slab_overhead" (12), "wgrad_reduce_cap" (8192)   split-K planning of the weight-gradient kernels
 *   "gn_fused" (1), "gn_bwd16" (1)   single-pass GroupNorm kernels; 512 x 16-pixel form of the 1024-pixel backward
 *   "attn_mfma" (1), "edge_mfma" (1) matrix-core attention (head dims 8, 16) and edge convolutions
 *   "timing_skip" (0)     TIMING EXPERIMENTS ONLY, results are wrong while non-zero: bit mask of kernel families whose
 *                         launches return without running (1: 1x1 convolutions, 2: weight-gradient slab sums, 4:
 *                         single-tap weight gradients, 8 / 16: GroupNorm backward / forward, 32: stride-2 convolutions,
 *                         64: three-tap weight gradients, 128: 3x3 stride-1 convolutions), to measure what a family
 *                         costs the step end to end
 * Unknown keys fail with DM_ERR_ARG. */
int dm_config_set(const char* key, int value);
int dm_config_get(const char* key, int* value);
/* every switch back to the value in brackets above (the configuration the library starts with) */
void dm_config_reset(void);

/* ---------------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on bf16 MFMA.
 * Replaces nn.Conv2d in modeling/blocks.py:26,28 (3x3 'same'), :45 (1x1 res_proj), :89 (3x3 stride 2 pad 1),
 * :110 (3x3 after F.interpolate, :116), the nn.Linear projections of modeling/attention.py:59-60 (as 1x1
 * convs over B*L "pixels"), torch.cat of modeling/blocks.py:290 (two-source channel axis), and their autograd
 * gradients.
 *
 * x1|x2 : bf16 [B][Hs][Ws][C1] and optionally [B][Hs][Ws][C2]; the conv sees channels (x1 | x2)
 * w      : bf16 [Cout][KH*KW][C1+C2]
 * ups=1  : the conv input is the nearest-neighbour x2 upsampling of the source (never materialised)
 * transposed=1 : data-gradient gather. Then x1 is dy [B][Hs][Ws][C1] of a forward conv with this
 *          KH/KW/stride/pad whose INPUT was [B][Ho][Wo][Cout]; w is [Cin_fwd][KH*KW][Cout_fwd] (channel roles
 *          swapped, taps NOT flipped); the result is dx.
 */
typedef struct dm_conv_desc {
  int B, Hs, Ws;
  int C1, C2;
  int Cout;
  int KH, KW, stride, pad;
  int ups;
  int transposed;
  int Ho, Wo;
} dm_conv_desc;

/* y[m][n] = conv(x)[m][n] + bias[n] + res[m][n]; bias (fp32 [Cout]) and res (bf16 [M][Cout]) may be null */
int dm_conv2d_igemm(const dm_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                    const void* res, void* y, void* stream);

/* The convolution and the GroupNorm (+ per-sample scale/shift or pre-add) (+ SiLU) that follows it in ONE launch
 * (modeling/blocks.py:51 -> :59-70: conv1, then norm2 mixed with the embedding, then the activation): on layers whose
 * 256-pixel output tiles hold whole samples and whole groups (images of at most 16x16), the statistics and the
 * normalised, activated tensor come out of the convolution's epilogue; y (the raw output, which the backward pass
 * needs) is written as by dm_conv2d_igemm, y_norm / mean / rstd as by dm_groupnorm_act_fwd on y (statistics taken from
 * the bf16-rounded y, pre_add folded in as there). g describes the GroupNorm over y (C1 = Cout, C2 = 0).
 * y may be null, and mean / rstd (both) as well: the raw output and the statistics are then not written - only a backward
 * pass reads them, so a forward pass without gradients (sampling, diffusion/sampler.py:52) saves that traffic.
 * dm_conv2d_gn_fusable tells whether a layer qualifies; dm_conv2d_igemm_gn fails with DM_ERR_UNSUPPORTED when it does not. */
struct dm_gn_desc;
int dm_conv2d_gn_fusable(const dm_conv_desc* d, const struct dm_gn_desc* g);
int dm_conv2d_igemm_gn(const dm_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                       const void* res, void* y, const struct dm_gn_desc* g, const float* gamma, const float* beta,
                       const float* scale_shift, const float* pre_add, void* y_norm, float* mean, float* rstd,
                       void* stream);

/* A 3x3 / stride-1 convolution and a 1x1 convolution of a SECOND input of the same spatial size, summed, in ONE launch:
 *   y = conv3x3(x1 | x2, w) + bias + conv1x1(p.x1 | p.x2, p.w) + p.bias
 * which is a ConvBlock's exit, `conv2(h) + res_proj(x)` (modeling/blocks.py:42-45, 73-75), when the block changes the
 * channel count: the projection's input chunks are multiplied into the same accumulator tile behind the 3x3 taps, so the
 * projected residual is never written to memory and read back. p.x1 | p.x2: bf16 [B][Ho][Wo][C1 | C2] (x2 null when
 * C2 == 0), p.w: bf16 [Cout][C1+C2], p.bias: fp32 [Cout] or null. dm_conv2d_proj_fusable tells whether the layer takes it
 * (3x3 / stride 1 / pad 1 layers of the patch kernel, no fused upsampling, C1 and C2 multiples of 64);
 * dm_conv2d_igemm_proj fails with DM_ERR_UNSUPPORTED when it does not. The gradients are those of the two convolutions
 * (dm_conv2d_igemm transposed, dm_conv2d_wgrad), one by one. */
typedef struct dm_conv_proj {
  const void* x1;
  const void* x2;
  const void* w;
  const float* bias;
  int C1, C2;
} dm_conv_proj;
int dm_conv2d_proj_fusable(const dm_conv_desc* d, const dm_conv_proj* p);
int dm_conv2d_igemm_proj(const dm_conv_desc* d, const void* x1, const void* x2, const void* w, const float* bias,
                         const dm_conv_proj* p, void* y, void* stream);

/* dw[n][tap][c] += sum_m dy[m][n] * X(m,tap)[c]; dw fp32 [Cout][KH*KW][C1+C2], ACCUMULATES.
 * ws: caller-provided scratch (16-byte aligned, at least dm_conv2d_wgrad_ws_bytes(d) bytes, contents irrelevant on
 * entry and on return). The split-K partial tiles go to the workspace with plain stores and a second kernel adds
 * their sum to dw in slice order: no float atomics, so the result does not depend on the order in which workgroups
 * finish (bitwise reproducible).
 * The weight gradient autograd computes for nn.Conv2d of modeling/blocks.py:26,28,45,89,110 inside loss.backward()
 * (train.py:440). */
size_t dm_conv2d_wgrad_ws_bytes(const dm_conv_desc* d);
int dm_conv2d_wgrad(const dm_conv_desc* d, const void* x1, const void* x2, const void* dy, float* dw, void* ws,
                    size_t ws_bytes, void* stream);

/* The weight gradients of several layers in one call (the same sums as n calls of dm_conv2d_wgrad, bit for bit per
 * layer given the same pixel split): runs of up to 8 consecutive 3x3 / stride-1 layers share one launch, so each layer
 * is cut into fewer, longer pixel slices and fewer partial tiles are written and summed. ws: 256-byte aligned scratch of
 * at least dm_conv2d_wgrad_group_ws_bytes(items, n) bytes. Nothing in the reference batches layers: autograd computes
 * these gradients layer by layer inside loss.backward() (train.py:440); they feed only the optimizer (train.py:447),
 * which is what allows a few layers to wait for each other. */
typedef struct {
  dm_conv_desc desc;
  const void* x1;
  const void* x2;   /* null when desc.C2 == 0 */
  const void* dy;
  float* dw;        /* fp32 [Cout][KH*KW][C1+C2], ACCUMULATES */
} dm_wgrad_item;
size_t dm_conv2d_wgrad_group_ws_bytes(const dm_wgrad_item* items, int n);
int dm_conv2d_wgrad_group(const dm_wgrad_item* items, int n, void* ws, size_t ws_bytes, void* stream);

/* fp32 parameter [Cout][Cin][KH][KW] with arbitrary element strides -> bf16 [Cout][T][Cin] (w_fwd) and
 * bf16 [Cin][T][Cout] (w_dgrad); either output may be null */
int dm_pack_conv_weight(const float* w, long s_co, long s_ci, long s_kh, long s_kw, int Cout, int Cin, int KH,
                        int KW, void* w_fwd, void* w_dgrad, void* stream);

/* Batched re-layout of bf16 conv/linear weights that live in one arena in the forward layout [Cout][T][Cin]
 * (the physical layout of the fp32 parameter arena, so the optimizer's bf16 copy IS the forward operand) into the
 * data-gradient layout [Cin][T][Cout] at the same element offsets of a second arena. `items_dev` is DEVICE memory
 * (n_items entries sorted by tile_begin; an item owns T * ceil(Cout/64) * ceil(Cin/64) tiles starting at tile_begin;
 * offsets are elements and multiples of 8, Cout and Cin multiples of 8), n_tiles the total. Replaces the per-layer
 * weight casts autocast performs in front of every conv/linear (modeling/blocks.py:51,70, attention.py:70,83). */
typedef struct dm_wt_item {
  long offset;
  int Cout, T, Cin, tile_begin;
} dm_wt_item;
int dm_transpose_conv_weights(const void* src, void* dst, const dm_wt_item* items_dev, int n_items, int n_tiles,
                              void* stream);

/* out[c] += sum_m x[m][c]  (bias gradients; x bf16 [M][C], out fp32). ws: scratch of dm_colsum_ws_floats(M, C)
 * floats; row chunks are summed into it with plain stores and added to out in chunk order (no float atomics).
 * The bias gradient autograd computes for nn.Conv2d (modeling/blocks.py:26,28,45,89,110). */
long dm_colsum_ws_floats(long M, int C);
int dm_colsum_bf16(const void* x, float* out, long M, int C, float* ws, dm_sum_item* deferred, void* stream);
/* the same sums added into two outputs: a block's conv2 and res_proj (modeling/blocks.py:28,45,74-75) produce the
 * two terms of one sum, so their bias gradients are the same column sums */
int dm_colsum_bf16_dual(const void* x, float* out, float* out2, long M, int C, float* ws, dm_sum_item* deferred,
                        void* stream);
/* out[c] += sum_r x[r][c]  (x fp32 [R][C]) */
int dm_colsum_f32(const float* x, float* out, int R, int C, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Edge convolutions (few channels on one side; HBM-bound, VALU).
 * in:  modeling/unet.py:44,147  (in_conv,  NCHW fp32 image -> NHWC bf16 features)
 * out: modeling/unet.py:126,162 (out_conv, NHWC bf16 features -> NCHW fp32 image)
 * Weights stay in the parameter layout [Cout][Cin][3][3] fp32 (contiguous).
 */
int dm_in_conv_fwd(const float* x, const float* w, const float* bias, void* y, int B, int Cs, int H, int W, int Cb,
                   void* stream);
int dm_out_conv_fwd(const void* h, const float* w, const float* bias, float* y, int B, int Cs, int H, int W, int Cb,
                    void* stream);
/* dh[m][cb] = sum_{co,tap} dy[b][co][...] * w[co][cb][tap]   (gradient w.r.t. out_conv's input) */
int dm_out_conv_dgrad(const float* dy, const float* w, void* dh, int B, int Cs, int H, int W, int Cb, void* stream);
/* Gradients of a 3x3 'same' edge conv's weight (and bias): `small` is the NCHW fp32 side with Cs channels, `big`
 * the NHWC bf16 side with Cb channels.
 *   big_is_output=1 (in_conv):  dw[cb][cs][kh][kw] += sum big[m][cb] * small[b][cs][h+kh-1][w+kw-1]; dbias[cb] += sum big
 *   big_is_output=0 (out_conv): dw[cs][cb][kh][kw] += sum small[b][cs][h][w] * big[(h+kh-1, w+kw-1)][cb]; dbias[cs] += sum small
 */
/* ws: scratch of dm_edge_conv_wgrad_ws_floats(...) floats: every workgroup leaves its partial sums there and they are
 * added to dw / dbias in workgroup order (no float atomics). */
long dm_edge_conv_wgrad_ws_floats(int B, int Cs, int H, int W, int Cb, int big_is_output);
int dm_edge_conv_wgrad(const float* small, const void* big, float* dw, float* dbias, int B, int Cs, int H, int W,
                       int Cb, int big_is_output, float* ws, dm_sum_item* deferred, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * GroupNorm (+ per-sample scale/shift or pre-add) (+ SiLU), fused; channel axis may be two sources.
 * Replaces nn.GroupNorm + nn.SiLU in modeling/blocks.py:50-51, :59-70, modeling/unet.py:160-161,
 * modeling/attention.py:67, and torch.cat of modeling/blocks.py:290.
 *   xin = (x1 | x2)[b][p][c] + pre_add[b][c]
 *   v   = gamma[c] * (xin - mean[b][g]) * rstd[b][g] + beta[c]
 *   u   = has_ss ? (1 + ss[b][c]) * v + ss[b][C + c] : v
 *   y   = act ? silu(u) : u                                   (bf16 [B][HW][C])
 */
typedef struct dm_gn_desc {
  int B, HW, C1, C2, G;
  float eps;
  int act;     /* 0 none, 1 SiLU */
  int has_ss;  /* scale_shift fp32 [B][2C] present */
  int has_pre; /* pre_add fp32 [B][C] present */
} dm_gn_desc;

/* fp32 elements of the caller-provided workspace `ws` both calls need (3*B*C). Pure scratch: every call writes it
 * before reading it, so one buffer can serve all calls issued on one stream. */
long dm_groupnorm_ws_floats(const dm_gn_desc* d);
int dm_groupnorm_act_fwd(const dm_gn_desc* d, const void* x1, const void* x2, const float* gamma, const float* beta,
                         const float* scale_shift, const float* pre_add, void* y, float* mean, float* rstd, float* ws,
                         void* stream);
/* The forward operator for callers that cannot allocate or pass a workspace (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): the same kernels with a fixed scratch array inside the library
 * standing in for `ws`. It has no counterpart in the reference. Nothing is allocated or freed, here or in any call
 * it makes. Limit: dm_groupnorm_ws_floats(d) <= dm_groupnorm_noalloc_max_ws_floats() (1 Mi floats = 4 MiB held per
 * GPU until the process ends; e.g. batch 256 x 1024 channels); larger shapes are rejected with DM_ERR_UNSUPPORTED.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race.
 * dm_noalloc_init() resolves the array's address for the CURRENT device (the first call of the operator on a device
 * does it otherwise); call it once per device before capturing a graph. */
int dm_noalloc_init(void);
long dm_groupnorm_noalloc_max_ws_floats(void);
int dm_groupnorm_act_fwd_noalloc(const dm_gn_desc* d, const void* x1, const void* x2, const float* gamma,
                                 const float* beta, const float* scale_shift, const float* pre_add, void* y,
                                 float* mean, float* rstd, void* stream);
/* dx1|dx2 bf16 (= GN backward + dextra + dextra2, both optional bf16 [B][HW][C]: gradients that reach the same input
 * by other routes - the residual branch of a block, a skip connection's second consumer - added here instead of by
 * a pass of their own; dextra2 only together with dextra); dscale_shift fp32 [B][2C] and dpre_add
 * fp32 [B][C] are written when present. param_partials (fp32 [B][3][C], written): every sample's contribution to
 * dbeta ([b][0][c]), dgamma ([b][1][c]) and, when want_upstream_bias (excludes dextra), the sum of dx over the
 * sample's pixels ([b][2][c]): the bias gradient of the convolution whose output x is (modeling/blocks.py:51 -> :59)
 * falls out of the sums this operator forms anyway, so that convolution needs no column-sum pass over dx.
 * dm_groupnorm_param_grads adds the batch sum of those terms, taken in sample order (no float atomics: bitwise
 * reproducible), to dgamma / dbeta / dupstream_bias (fp32 [C], ACCUMULATE; dupstream_bias nullable). It may run on
 * another stream than the backward kernel, once that has finished: nothing but the optimizer consumes its results. */
int dm_groupnorm_act_bwd(const dm_gn_desc* d, const void* x1, const void* x2, const float* gamma, const float* beta,
                         const float* scale_shift, const float* pre_add, const float* mean, const float* rstd,
                         const void* dy, const void* dextra, const void* dextra2, void* dx1, void* dx2,
                         float* dscale_shift, float* dpre_add, float* param_partials, int want_upstream_bias,
                         float* ws, void* stream);
int dm_groupnorm_param_grads(const dm_gn_desc* d, const float* param_partials, float* dgamma, float* dbeta,
                             float* dupstream_bias, dm_sum_item* deferred, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Spatial self-attention core. Replaces F.scaled_dot_product_attention and the view/permute/contiguous
 * shuffles of modeling/attention.py:71-79.
 * qkv bf16 [B][L][3][H][D] (the qkv_proj output as is), out bf16 [B][L][H][D], lse fp32 [B][H][L]; D in {8,16,32,64}
 */
int dm_attention_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int D, float scale, void* stream);
int dm_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int L,
                     int H, int D, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Embedding path (fp32, tiny): modeling/embeddings.py:24-27 (sinusoidal encoding), :43-47 (MLP),
 * modeling/blocks.py:56-57 (emb_proj of SiLU(embedding)), modeling/unet.py:138-145 (class embedding add).
 */
int dm_timestep_embed(const int64_t* t, const float* freq, float* out, int B, int half, void* stream);
/* y[m][n] = sum_k f(x[m][k]) * w[n][k] + bias[n], f = SiLU if act_in else identity */
int dm_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act_in,
                  void* stream);
/* dx[m][k] = f'(x[m][k]) * sum_n dy[m][n] w[n][k]  (dx may be null);  dw[n][k] += sum_m dy[m][n] f(x[m][k]);
 * dbias[n] += sum_m dy[m][n]  (dw/dbias may be null) */
int dm_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* dbias, int M, int K,
                  int N, int act_in, void* stream);
/* P <= 32 layers sharing one input x [M][K] in ONE launch (the per-block emb_proj layers, modeling/blocks.py:56-57,
 * all read SiLU(embedding)). `items` is a HOST array; the pointers inside are device pointers.
 * forward:  y_p[m][n] = sum_k f(x[m][k]) w_p[n][k] + b_p[n]
 * backward: dx[m][k] = f'(x[m][k]) * sum_p sum_n dy_p[m][n] w_p[n][k]  (dx may be null);
 *           dw_p += dy_p^T f(x), db_p += column sums of dy_p  (accumulate; either may be null) */
typedef struct dm_linear_item {
  const float* w;   /* [N][K] */
  const float* b;   /* [N] or null */
  float* y;         /* [M][N]   (forward) */
  const float* dy;  /* [M][N]   (backward) */
  float* dw;        /* [N][K]   (backward, accumulates) */
  float* db;        /* [N]      (backward, accumulates) */
  int N;
} dm_linear_item;
int dm_linear_group_fwd(const float* x, const dm_linear_item* items, int P, int M, int K, int act_in, void* stream);
/* ws: scratch of dm_linear_group_bwd_ws_floats(items, P, M, K) floats (16-byte aligned; needed when dx is given): the
 * input gradient is summed from per-slice partial tiles in a fixed order, without float atomics */
long dm_linear_group_bwd_ws_floats(const dm_linear_item* items, int P, int M, int K);
int dm_linear_group_bwd(const float* x, const dm_linear_item* items, int P, float* dx, float* ws, int M, int K,
                        int act_in, void* stream);
/* emb[b][:] += table[idx(b)][:], idx(b) = (cls == null || mask[b] != 0) ? uncond_idx : cls[b] */
int dm_class_embed_add(float* emb, const float* table, const int64_t* cls, const float* mask, int uncond_idx, int B,
                       int E, void* stream);
/* dtable[idx(b)][:] += demb[b][:] for idx(b) != uncond_idx (the padding row never receives gradient); dtable has
 * n_rows rows; the samples of a row are added in batch order */
int dm_class_embed_bwd(const float* demb, float* dtable, const int64_t* cls, const float* mask, int uncond_idx,
                       int n_rows, int B, int E, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Resampling without conv (modeling/blocks.py:92 AvgPool2d, :116 F.interpolate nearest) and the 2x2 gather of
 * the fused-upsample conv's data gradient. All bf16 NHWC.
 */
int dm_upsample2x_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream);  /* y [B][2H][2W][C] */
int dm_upsample2x_bwd(const void* dy, void* dx, int B, int H, int W, int C, void* stream); /* dx[h][w] = sum of 4 */
int dm_avgpool2x_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream);   /* y [B][H/2][W/2][C] */
int dm_avgpool2x_bwd(const void* dy, void* dx, int B, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Diffusion process (all fp32, NCHW or any layout with the sample index slowest; n = elements per sample).
 * mode 0 (variance preserving, diffusion/scheduler.py:75-83): a = (float)sqrt(tab64[t]), s = (float)sqrt(1 - tab64[t])
 * mode 1 (rectified flow,      diffusion/scheduler.py:149-153): s = tab32[t], a = 1 - s
 *   x_t = a * x + s * eps
 */
int dm_diffuse(const float* x, const float* eps, const int64_t* t, const double* tab64, const float* tab32,
               float* x_t, int B, long n, int mode, void* stream);
/* loss = mean((pred - target)^2), target = eps (mode 0, scheduler.py:72) or x - eps (mode 1, scheduler.py:146).
 * partials: fp32 scratch of dm_mse_partials() elements; loss: one fp32 */
int dm_mse_partials(void);
int dm_mse_loss_fwd(const float* pred, const float* x, const float* eps, float* partials, float* loss, long total,
                    int mode, void* stream);
/* dpred = gout[0] * 2 (pred - target) / total */
int dm_mse_loss_bwd(const float* pred, const float* x, const float* eps, const float* gout, float* dpred, long total,
                    int mode, void* stream);

/* One sampler update, in place on z (diffusion/sampler.py). Scalars are the per-step table entries.
 * mode 0 DDPM  (:80-84):    z = c0 * (z - c1 * pred);  if noise: z += c2 * noise
 * mode 1 DDIM  (:116-130):  x0 = (z - c1 * pred) / c0; if clip: clamp(x0,-1,1); z = last ? x0 : c2 * x0 + c3 * pred
 * mode 2 Euler (:157-162):  z = z - c0 * pred
 */
int dm_sampler_step(float* z, const float* pred, const float* noise, float c0, float c1, float c2, float c3, int mode,
                    int clip, int last, long total, void* stream);
/* The same update for loop step i = step[0] (device counter), its scalars (c0..c3) read from row i of the device table
 * `coefs` [n_steps][4]; last = (i == n_steps-1), on which `noise` is ignored. Afterwards step[0] += 1 and
 * t_out[0..B) = t_table[step[0]] (clamped to the last entry): the timestep argument of the next UNet forward. Nothing in
 * the launch depends on the step index, so ONE captured HIP graph (UNet forward + this call) replays a whole sampling
 * loop (diffusion/sampler.py:76-84, 111-130, 152-162) without per-step host work. Replays beyond n_steps change nothing. */
int dm_sampler_step_dev(float* z, const float* pred, const float* noise, const float* coefs, long long* step,
                        const int64_t* t_table, int64_t* t_out, int n_steps, int B, int mode, int clip, long total,
                        void* stream);
/* out = u + s * (c - u)   (modeling/predictor.py:29) */
int dm_cfg_combine(const float* pred_uncond, const float* pred_cond, float* out, float s, long total, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Optimizer tail on flat fp32 arenas (train.py:441-450: clip_grad_norm_, AdamW.step, EMAModel.step).
 */
int dm_sumsq_partials(void);
/* norm_out[0] = |grad_scale| * sqrt(sum g^2): the global L2 norm of grad_scale * g (clip_grad_norm_'s norm,
 * train.py:441-444). grad_scale is 1 on one GPU and 1 / world size when g holds the SUM of the ranks' gradients (the
 * data-parallel mean is then never written to memory). partials: fp32 scratch of dm_sumsq_partials() elements */
int dm_grad_norm(const float* g, long n, float grad_scale, float* partials, float* norm_out, void* stream);
/* The gradient used is grad_scale * g, with clip = min(1, max_norm / (norm[0] + 1e-6)) applied on the fly (max_norm <= 0:
 * no clipping; norm as dm_grad_norm produced it with the same grad_scale);
 * AdamW with decoupled weight decay and bias corrections bc1 = 1 - b1^step, bc2 = 1 - b2^step; then
 * ema -= (1 - ema_decay) * (ema - p) when ema != null; w_bf16 (nullable) receives the bf16 copy of p.
 * zero_grad != 0: g is zeroed behind the read (optimizer.zero_grad of train.py:448 without a pass of its own). */
int dm_adamw_ema_step(float* p, float* g, float* m, float* v, float* ema, void* w_bf16, const float* norm,
                      float max_norm, long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                      float ema_decay, const float* hyper, float grad_scale, int zero_grad, void* stream);
/* Device-side step bookkeeping for graph replay: step[0] += 1; hyper = {lr, 1 - b1^step, 1 - b2^step, ema_decay}
 * with the warm-up EMA rule decay = min(1 - (1 + (step-1)/inv_gamma)^-power, max_decay) (0 on the first step).
 * When `hyper` (device, 4 floats) is passed to dm_adamw_ema_step it overrides lr, bc1, bc2 and ema_decay. */
int dm_optimizer_tick(long long* step, float* hyper, float lr, float b1, float b2, float ema_inv_gamma, float ema_power,
                      float ema_max_decay, void* stream);
/* The same with a learning-rate schedule evaluated on the device, so that a replayed HIP graph follows it:
 * hyper[0] = base_lr * multiplier(step - 1). Replaces lr_scheduler.step() of train.py:446-447 / :281-291
 * (transformers.get_scheduler -> torch LambdaLR; optimizer step k, 0-based, runs at base_lr * lambda(k)).
 * kind: the DM_LR_* rules below, formulas of transformers/optimization.py. warmup / total = num_warmup_steps /
 * num_training_steps. p0: num_cycles (cosine 0.5, cosine_with_restarts 1), power (polynomial), timescale
 * (inverse_sqrt); p1: lr_end / lr_init (polynomial). dm_lr_multiplier evaluates the same rule on the host. */
enum {
  DM_LR_CONSTANT = 0,
  DM_LR_CONSTANT_WITH_WARMUP = 1,
  DM_LR_LINEAR = 2,
  DM_LR_COSINE = 3,
  DM_LR_COSINE_RESTARTS = 4,
  DM_LR_POLYNOMIAL = 5,
  DM_LR_INVERSE_SQRT = 6
};
int dm_optimizer_tick_lr(long long* step, float* hyper, float base_lr, int kind, long long warmup, long long total,
                         double p0, double p1, float b1, float b2, float ema_inv_gamma, float ema_power,
                         float ema_max_decay, void* stream);
double dm_lr_multiplier(int kind, long long k, long long warmup, long long total, double p0, double p1);

/* ---------------------------------------------------------------------------------------------------------
 * Either side of the path: the data set resident in HBM, and the memorisation metric of the evaluation.
 */
/* A training batch from a uint8 image set in device memory, data [n_items][H][W][C] (the layout of torchvision's
 * CIFAR10.data; MNIST is C = 1): out[b][c][h][w] (fp32, NCHW) = normalise(data[idx[b]][h'][w'][c]), h'/w' mirrored
 * when flip[b] has bit 1 / bit 0 set (flip and idx nullable: no flips / rows 0..B-1).
 *   mode 0: ((v / 255) - 0.5) / 0.5  - ToTensor + Normalize(0.5, 0.5), the DataLoader transform (train.py:111-114)
 *   mode 1: v / 127.5 - 1            - the data_tensor form (train.py:124-125, 136)
 * each a separately rounded fp32 operation, bit-identical to torch. Replaces DataLoader + transforms
 * (train.py:312-329, utils/augmentation.py:6-18) for data sets that fit HBM. */
int dm_u8_batch(const uint8_t* data, const int64_t* idx, const uint8_t* flip, float* out, int B, int C, int H, int W,
                long n_items, int mode, void* stream);
/* d1[i], d2[i] = the two smallest L2 distances from generated row i (fp32 [n_gen][D]) to the reference rows (fp32
 * [n_ref][D], n_ref >= 2): torch.cdist(gen, ref).topk(2, largest=False) of utils/validate_memo.py:13-18, as an fp32
 * MFMA GEMM with the selection fused in (no [n_gen][n_ref] matrix is materialised).
 * ws: scratch of dm_nn2_ws_floats(n_gen, n_ref) floats. */
long dm_nn2_ws_floats(int n_gen, int n_ref);
int dm_nn2_distances(const float* gen, const float* ref, float* d1, float* d2, int n_gen, int n_ref, int D, float* ws,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif
