// Error reporting shared by every entry point of libdm_hip.so.
#include "dm_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void dm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dm_last_error(void) { return g_err; }
extern "C" int dm_abi_version(void) { return 3; }

// run-time switches (tests exercise both convolution paths)
int g_dm_use_patch_kernel = 1;
int g_dm_use_wgrad3 = 1;
int g_dm_wgrad_slab_overhead = 12;   // fixed cost of a weight-gradient workgroup, in 64-pixel steps, with slab output
int g_dm_wgrad_reduce_cap = 8192;    // most workgroups of a slab-sum kernel (latency-bound: more is better up to here)
int g_dm_gn_fused = 1;
int g_dm_use_wgrad9 = 3;              // 3x3 weight gradients with all nine taps per workgroup: 0 never, 1 from 512 input channels, 3 also from 256 on images of >= 16x16 (where it is 7-19 % faster than the three-tap kernel alone; the step is the same within 0.1 % either way), 2 wherever the shape allows
int g_dm_dgrad_subpixel = 1;          // stride-2 data gradients by output-pixel parity class (only the taps a class meets)
int g_dm_conv_gn_fuse = 1;            // GroupNorm(+SiLU) that follows a 3x3 conv in that conv's epilogue where a tile holds whole samples
int g_dm_conv_proj_fuse = 1;          // a ConvBlock's 1x1 residual projection inside its second 3x3 convolution's launch
int g_dm_edge_mfma = 1;               // UNet edge convs (image <-> features): matrix-core kernels instead of the VALU ones
int g_dm_attn_mfma = 1;              // head dims 8 / 16: attention core on the matrix cores
// TIMING EXPERIMENTS ONLY (results are wrong while set): a bit mask of launches that return at once without running,
// to measure what a family of kernels costs the step end to end (what an infinitely fast replacement would buy):
// 1: 1x1 convolutions (forward / data gradient), 2: the weight gradients' slab-sum kernels, 4: single-tap weight
// gradients (1x1, stride 2), 8: GroupNorm backward, 16: GroupNorm forward, 32: stride-2 convolutions (forward / data
// gradient), 64: three-tap weight gradients, 128: 3x3 stride-1 convolutions (forward / data gradient)
int g_dm_timing_skip = 0;
// The same families, launched TWICE (same bits; the results stay valid except that weight gradients whose slab sums
// run twice come out doubled): the step's extra time is what the family costs end to end, on real data.
int g_dm_timing_twice = 0;
int g_dm_conv1x1_stream = 1;          // 1x1 stride-1 convolutions: streaming kernel (activations straight into MFMA operands) instead of the gathered one where it is faster (2: wherever the shape allows, 0: never)
int g_dm_patch_direct = 1;           // 3x3 patch kernel: plain outputs leave the accumulator registers directly (0: through LDS, as the outputs with a GroupNorm epilogue do)
int g_dm_gn_min_ppt = 4;             // single-pass GroupNorm: least pixels per thread (1, 2, 4, 8) where the image leaves 8 pixel lanes; 8x8 images: 128-thread workgroups of 4 pixels per thread instead of 512 x 1 (forward 10.0 -> 7.0 us, backward 14.1 -> 9.2 us at batch 128, 512 channels)
int g_dm_gn_bwd16 = 1;               // 513..1024-pixel GroupNorm backward: 512 threads x 16 pixels (no spills) instead of 1024 x 8

namespace {
struct Switch { const char* key; int* value; int initial; };
// every switch with the value the library starts with (dm_config_reset restores these)
const Switch kSwitches[] = {
  {"conv_patch", &g_dm_use_patch_kernel, 1},     {"wgrad_row3", &g_dm_use_wgrad3, 1},
  {"wgrad_slab_overhead", &g_dm_wgrad_slab_overhead, 12}, {"wgrad_reduce_cap", &g_dm_wgrad_reduce_cap, 8192},
  {"gn_fused", &g_dm_gn_fused, 1},               {"conv_gn_fuse", &g_dm_conv_gn_fuse, 1},
  {"wgrad_row9", &g_dm_use_wgrad9, 3},           {"dgrad_subpixel", &g_dm_dgrad_subpixel, 1},
  {"edge_mfma", &g_dm_edge_mfma, 1},             {"attn_mfma", &g_dm_attn_mfma, 1},
  {"timing_skip", &g_dm_timing_skip, 0},         {"timing_twice", &g_dm_timing_twice, 0},
  {"conv1x1_stream", &g_dm_conv1x1_stream, 1},   {"patch_direct", &g_dm_patch_direct, 1},
  {"gn_min_ppt", &g_dm_gn_min_ppt, 4},
  {"gn_bwd16", &g_dm_gn_bwd16, 1},               {"conv_proj_fuse", &g_dm_conv_proj_fuse, 1},
};
const Switch* find_switch(const char* key) {
  if (key)
    for (const Switch& s : kSwitches)
      if (!strcmp(key, s.key)) return &s;
  dm_set_error("dm_config: unknown key '%s'", key ? key : "(null)");
  return nullptr;
}
}  // namespace

extern "C" int dm_config_set(const char* key, int value) {
  const Switch* s = find_switch(key);
  if (!s) return DM_ERR_ARG;
  if (s->value == &g_dm_wgrad_reduce_cap && value < 1) value = 1;
  *s->value = value;
  return DM_OK;
}

extern "C" int dm_config_get(const char* key, int* value) {
  const Switch* s = find_switch(key);
  if (!s) return DM_ERR_ARG;
  DM_REQUIRE(value, DM_ERR_ARG, "dm_config_get: null output");
  *value = *s->value;
  return DM_OK;
}

extern "C" void dm_config_reset(void) {
  for (const Switch& s : kSwitches) *s.value = s.initial;
}
