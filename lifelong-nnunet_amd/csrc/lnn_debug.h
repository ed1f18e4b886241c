/* Debug / test hooks of liblnn_hip.so -- NOT part of the drop-in surface (include/lnn_hip.h).  The parity tests use them to pin
 * one kernel variant at a time, the profiling tools to read per-phase cycle counters.  Process-wide, not thread-safe. */
#ifndef LNN_DEBUG_H
#define LNN_DEBUG_H
#include "../../include/lnn_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* debug: every lane of one wave issues ds_read_b64_tr_b16 at LDS byte address lane*8 over an LDS image
 * holding its own half-index; out[lane*4+j] (float) = value received.  Used by tests to pin the
 * hardware transpose-read lane mapping the wgrad kernels rely on. */
int lnn_debug_tr16_probe(lnn_stream_t s, float* out256);

/* debug: when set to a zeroed device buffer of 6 uint64, the stride-1 conv kernel accumulates shader-clock cycles
 * per phase {issue loads, MFMA, barrier, LDS stores, barrier} and the step count; pass NULL to disable. */
int lnn_debug_set_phase_buffer(void* dev_ptr_6x_u64);
/* Parity tests only: pin the stride-1 conv forward / dgrad kernel (-1 automatic, 5 = the tile kernel "v5", 9 = the z-streaming kernel
 * where the layer has 32 / 64 / 128 input channels and no accumulation, 10 = the macro-tile kernel where the contraction is a multiple
 * of 16 channels and the band's halo fits its LDS image, the tile kernel otherwise; any other value is an error -- v7 / v8 were retired in
 * round 6).  Process-wide, not thread-safe: a debug hook, not part of the production surface. */
int lnn_debug_force_conv_kernel(int which);
/* Parity tests only: pin the stride-2 conv forward kernel (-1 automatic, 0 the tile kernel, 1 the z-streaming kernel wherever
 * it supports the layer: 32 / 64 input channels, output channels a multiple of 64, even extents).  Process-wide. */
int lnn_debug_force_down2_kernel(int which);
/* Parity tests only: number of z segments the v9 kernel cuts a column into (0 = automatic).  Process-wide. */
int lnn_debug_set_v9_zseg(int segments);

/* Parity tests only: [1,3,3] stride-1 convolutions (lnn_conv3d_fwd_g / lnn_conv3d_dgrad_g) on the z-streaming kernel with permuted
 * axes: -1 automatic (32 / 64 / 128 gathered channels, H >= 32), 0 never (the flattened-voxel kernel), 1 wherever the kernel supports
 * the shape; lnn_debug_last_k133_on_v9: 1 when the last such call ran on the z-streaming kernel.  Process-wide. */
int lnn_debug_set_k133_v9(int mode);
int lnn_debug_last_k133_on_v9(void);

/* Parity tests only: the isotropic entry points hand small volumes (<= 4096 output voxels) to the generic flattened-voxel
 * kernels of igemm_gen.hip; -1 = that automatic rule, 0 = never (pins the specialised kernels on the tests' small shapes),
 * 1 = every layer the generic kernels support.  Process-wide. */
int lnn_debug_set_gen_mode(int mode);

/* Parity tests only: 1 when the last lnn_conv3d_dgrad_in_bwd_sums call ran the fused epilogue (z-streaming kernel, EPI = 2), 0 when it
 * ran the two separate calls. */
int lnn_debug_last_dgrad_reduce_fused(void);

/* Measurements only: the persistent MFMA kernels size their grids for `cus` CUs instead of the whole device (0 = all) -- read at
 * launch time, so set / launch / reset brackets individual launches (tools/, profiles/r04_overlap_probe.txt).  Process-wide. */
int lnn_debug_set_cu_budget(int cus);

#ifdef __cplusplus
}
#endif
#endif /* LNN_DEBUG_H */
