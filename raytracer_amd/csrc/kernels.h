// Host-callable launchers of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "device_types.h"

namespace rtamd {

// `depth` = recursion depth of the stage: selects the ray queue (depth % 3) and the hit list (depth & 1).
// Start of a pass: zeroes `zero_bytes` (a multiple of 16, 16-byte aligned) of counters and fills the pass's screen-coordinate tables.
void launch_prep(const FrameConst &fc, const Workspace &ws, void *zero, size_t zero_bytes, void *zero_b, size_t zero_b_bytes, float *rgb,
                 int32_t *face_id, float *t_hit, hipStream_t stream);
// Primary rays of one pass resolved face by face into ws.pix_key (zeroed by launch_prep); launch_compact reads it.
// With ws.prefilled and ws.prefill_with_faces the launch also writes the miss values into the output planes (extra workgroups).
void launch_primary_faces(bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int num_tris, float *rgb, int32_t *face_id,
                          float *t_hit, hipStream_t stream);
// Screen binning of the primary rays of one pass (ws.bin_entries must be set): count, claim, fill, test.
void launch_primary_bins(bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int num_tris, bool cleared,
                         hipStream_t stream);
// `expect` (launches whose size is only known on the device): the count an earlier frame of the view reported for this
// point of the frame -- bounce rays of the stage, or its hits -- or -1: cover the worst case.
void launch_walk_closest(bool primary, bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth,
                         int expect, hipStream_t stream);
void launch_compact(bool primary, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, bool spawn,
                    bool list_misses, float *rgb, int32_t *face_id, float *t_hit, int expect, hipStream_t stream, int expect_hits = -1);
void launch_walk_shadow(bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, int expect,
                        hipStream_t stream);
void launch_shade(bool primary, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, bool spawn, float *rgb,
                  int expect, hipStream_t stream);
// One launch for two independent pieces of a frame (see shadow_walk_kernel, shade_compact_kernel in kernels.hip): the shadow
// rays of level `depth` with the closest hits of level depth + 1, and the shading of level `depth` with the compaction of
// level depth + 1.  `stage` is the stage of level `depth`; the other piece works on stage + 1.
bool can_fuse_shadow_walk(const DeviceScene &sc, const FrameConst &fc);
bool can_fuse_shade_compact(const Workspace &ws);
void launch_shadow_walk(bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, int expect_hits,
                        int expect_next, hipStream_t stream);
void launch_shade_compact(bool primary, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, bool spawn_next,
                          bool list_misses, float *rgb, int expect_hits, int expect_next, hipStream_t stream);
void launch_misses(const FrameConst &fc, const Workspace &ws, int stage, int depth, float *rgb, int expect, hipStream_t stream);
void launch_scale(float *rgb, size_t n, float divisor, hipStream_t stream);
void launch_quantize(const float *rgb, uint8_t *out, size_t n, hipStream_t stream);
void launch_quantize_checked(const float *rgb, uint8_t *out, size_t n, int *inexact, hipStream_t stream);
void launch_assemble(const void *src, void *dst, int width, int height, int channels, int row_tile, int num_parts, hipStream_t stream);

// Smallest widening (pixels) of projected boxes and triangles; see the error bound at RT_BIN_MARGIN in kernels.hip.
float raster_margin_floor();

} // namespace rtamd
