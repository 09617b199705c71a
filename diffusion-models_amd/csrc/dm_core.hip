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
extern "C" int dm_abi_version(void) { return 2; }

// run-time switches (tests exercise both convolution paths)
int g_dm_use_patch_kernel = 1;
int g_dm_use_wgrad3 = 1;
int g_dm_wgrad_slab_overhead = 12;   // fixed cost of a weight-gradient workgroup, in 64-pixel steps, with slab output
int g_dm_wgrad_reduce_cap = 8192;    // most workgroups of a slab-sum kernel (latency-bound: more is better up to here)
int g_dm_gn_fused = 1;
int g_dm_use_wgrad9 = 0;              // 3x3 weight gradients with all nine taps per workgroup (1: >= 512 input channels, 3: also >= 256 on >= 16x16 images, 2: wherever the shape allows). Off: faster alone, but the step is slower (its 236 VGPRs leave no room on a SIMD for the other stream's GroupNorm waves)
int g_dm_dgrad_subpixel = 1;          // stride-2 data gradients by output-pixel parity class (only the taps a class meets)
int g_dm_conv_gn_fuse = 1;            // GroupNorm(+SiLU) that follows a 3x3 conv in that conv's epilogue where a tile holds whole samples
int g_dm_edge_mfma = 1;               // UNet edge convs (image <-> features): matrix-core kernels instead of the VALU ones
int g_dm_attn_mfma = 1;              // head dims 8 / 16: attention core on the matrix cores
int g_dm_gn_bwd16 = 1;               // 513..1024-pixel GroupNorm backward: 512 threads x 16 pixels (no spills) instead of 1024 x 8

extern "C" int dm_config_set(const char* key, int value) {
  if (key && !strcmp(key, "conv_patch")) { g_dm_use_patch_kernel = value; return DM_OK; }
  if (key && !strcmp(key, "wgrad_row3")) { g_dm_use_wgrad3 = value; return DM_OK; }
  if (key && !strcmp(key, "wgrad_slab_overhead")) { g_dm_wgrad_slab_overhead = value; return DM_OK; }
  if (key && !strcmp(key, "wgrad_reduce_cap")) { g_dm_wgrad_reduce_cap = value > 0 ? value : 1; return DM_OK; }
  if (key && !strcmp(key, "gn_fused")) { g_dm_gn_fused = value; return DM_OK; }
  if (key && !strcmp(key, "conv_gn_fuse")) { g_dm_conv_gn_fuse = value; return DM_OK; }
  if (key && !strcmp(key, "wgrad_row9")) { g_dm_use_wgrad9 = value; return DM_OK; }
  if (key && !strcmp(key, "dgrad_subpixel")) { g_dm_dgrad_subpixel = value; return DM_OK; }
  if (key && !strcmp(key, "edge_mfma")) { g_dm_edge_mfma = value; return DM_OK; }
  if (key && !strcmp(key, "attn_mfma")) { g_dm_attn_mfma = value; return DM_OK; }
  if (key && !strcmp(key, "gn_bwd16")) { g_dm_gn_bwd16 = value; return DM_OK; }
  dm_set_error("dm_config_set: unknown key '%s'", key ? key : "(null)");
  return DM_ERR_ARG;
}
