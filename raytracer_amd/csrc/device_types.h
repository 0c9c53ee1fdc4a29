// Plain structs shared by the host orchestration (api.cpp) and the gfx950 kernels (kernels.hip).
#pragma once
#include <cstdint>

namespace rtamd {

constexpr int kStackEntries = 64;   // per-lane traversal stack (LDS + private memory); scene creation checks the tree against it
constexpr int kMaxShadowRays = 512; // lights * samples_per_light per hit
constexpr int kShadeRecordQuads = 8; // float4 per ShadeRecord (bvh.hpp)
constexpr int kMaterialFloats = 12;  // ka, kd, ks, shininess, texture index (int bits), pad
constexpr int kMaxStages = 64;      // (ssaa passes) * (max_reflections + 1) counter slots per frame

// Read-only scene in HBM.
struct DeviceScene {
    const float4 *nodes;      // 8 x float4 per Bvh4Node
    const float4 *tris;       // 4 x float4 per TriRecord
    const float4 *planes;     // (n, D) of every TriRecord once more, densely packed: what a test reads before it needs the corners
    const float4 *shade;      // 8 x float4 per ShadeRecord, indexed like tris
    const float *materials;   // 12 floats per material: ka, kd, ks, shininess, texture index (int bits), pad
    const float *texels;      // all textures back to back, 3 floats per texel
    const int32_t *tex_info;  // per texture: first texel, width, height, 0
    const float *light_pos;   // 3 per light
    const float *samples;     // 3 per (light, sample)
    const float4 *spheres;    // center, radius
    const int32_t *tri_of_rank; // record index of the face with a given rank (position in the reference's visiting order), -1 = none
    const float4 *cluster_box;  // per 64 consecutive face records: lower corner, upper corner of their vertices (two float4)
    int32_t num_lights, spl, num_spheres, num_materials; // num_lights: lights WITH shadow sample points (getLightIntensity's loop)
    int32_t num_shade_lights; // entries of light_pos (shadeOffFace's loop): num_lights + the lights added after the samples were made
    int32_t num_textures; // 0 also when the mesh has no texture coordinates
    // shadow-ray candidate grids (shadow_grid.hpp), one per light sample; use_grids = 0: shadow rays walk the tree
    const float4 *grid_headers;   // 4 x float4 per ShadowGridHeader
    const int32_t *grid_cells;    // cell-start array of all grids
    const int2 *grid_entries;     // (TriRecord index, zmax bits)
    int32_t use_grids;
    // the (up to) sixteen grandchildren of the root, where the lanes of a shared ray start (0x7fffffff: none)
    int32_t start16[16];
    int32_t num_records; // TriRecords (= planes = shading records)
    // Slack of the tree's boxes in space: 2^-20 of the largest coordinate of any face or of the ray's origin, whichever is larger
    // (box_slack_rel = 2^-20).  The exact face test rounds, so it may accept a point a few units in the last place of those
    // coordinates outside a face's own box (a ray through a corner of the face, a ray that runs IN the plane of a box side); the
    // slab tests let a ray pass that misses a box by less than this.
    float box_slack, box_slack_rel;
    float largest_coord; // largest coordinate of any face
};

// Per-launch constants (camera + the reference's macros).
struct FrameConst {
    float origin[3];
    float inv_view[12];
    float vp[4];
    float ax;    // aspect_ratio * scale   (camera.hpp:169)
    float scale; // 1 / perspective scale  (camera.hpp:168)
    float sub_x, sub_y; // sub-sample offset of this SSAA pass (src/flyscene.cpp:816-817)
    int32_t width, height, local_rows;
    int32_t tiles_x, tiles_y; // 8x8-pixel tiles of the local image
    int32_t row_tile, num_parts, part;
    uint32_t num_parts_magic; // floor(2^32 / num_parts), likewise
    uint32_t tiles_x_magic, row_tile_magic; // floor(2^32 / tiles_x), floor(2^32 / row_tile) (0xffffffff for 1): fast_div in kernels.hip
    float t_eps, refl_thresh, bounce_off;
    float bg[3], bg_deep[3];
    int32_t accumulate; // 0: first SSAA pass stores the pixel, 1: later passes add to it
    // screen binning of the primary rays (raster_kernel): inverse of the linear part of inv_view, and that
    // inverse applied to (inv_view translation - origin); bins of kBinPixels^2 pixels over the whole image
    float lin_inv[9];
    float cam_e[3];
    float proj_s;                               // 1 / (cam_e[2] - 1)
    float px_off, px_scale, py_off, py_scale;   // pixel = (px_off + n.x * px_scale, py_off - n.y * py_scale) for a direction n of the camera
    int32_t bins_x, bins_y;
    float raster_margin; // widening of projected boxes and triangles in pixels (error bound: RT_BIN_MARGIN in kernels.hip)
    // ... plus coord_slack times a corner's spread (project_point): the exact test computes with the coordinates themselves, so its
    // decisions are only good to a few units in the last place of the LARGEST coordinate involved (eye or face), which is more
    // pixels the nearer the face is to the eye: coord_slack = 2^-19 * that coordinate * the projection's pixels per unit of n
    float coord_slack;
};

constexpr int kBinPixels = 4;   // screen bin edge; an 8x8 slot tile holds four bins
constexpr int kBigBins = 256;   // a face covering more bins goes to the list every pixel scans

// State of the per-pass screen binning, zeroed with the stage counters.
struct BinState {
    unsigned long long total; // entries asked for so far (bin ranges are claimed block by block); 64 bits: a face may enter 256 bins
    int32_t num_big;  // faces in the big list
    int32_t fallback; // 1: the entry buffer is too small for this view -> the primary rays walk the tree
    int32_t nonempty; // bins that hold at least one face
    int32_t pad;
};

// Longest average list (entries per non-empty bin) for which the bins beat the tree: every pixel of a bin
// runs through the whole list, one batch of four after the other, while a tree walk meets 10..35 boxes and a
// few faces whatever the image size.  Measured crossover (69 K-face and 1.1 M-face scenes, 480x270 .. 4K):
// bins win at 21 and 32 entries per bin, the tree at 46 and beyond.
constexpr int kMaxAverageBinList = 42;

// Per-stage counters (one stage = one recursion level of one SSAA pass), zeroed once per frame.
struct StageCounters {
    int32_t hits;   // entries of the hit list (written by the compaction kernel)
    int32_t next;   // rays appended to the next queue (written by the compaction kernel)
    int32_t misses; // bounce rays that hit nothing, listed only when a deep background colour is set
    int32_t pad;
};

struct WorkCounters {
    unsigned long long box_tests, tri_plane, tri_edge, pad;
};

// Shadow rays actually shot in a frame: every wave of a shadow-ray kernel adds the rays it processed, once, to one of
// kShotSlots counters, each on a 64-byte line of its own (one counter would take tens of thousands of atomics on one
// address per launch, which serialise); the host adds the slots up.
constexpr int kShotSlots = 64;
constexpr int kShotStride = 8; // 64-bit words between two slots
constexpr int kCounterHeadBytes = 8192; // stage counters, work counters, shot counters: zeroed at the start of a frame

// Hit list of one stage: positions 0..hits-1, written by the compaction kernel.  Two lists alternate by
// recursion depth so that the lighting of depth d can run while depth d+1 is traced.
struct HitList {
    int32_t *slot;  // ray slot: tile-ordered pixel (primary) or queue index (bounce); misses are listed from the back
    int32_t *tri;   // TriRecord index, -1 sphere
    int32_t *child; // index of the reflection ray in the next queue, -1 none
    float *p;       // 3 planes of `capacity`: hit point
};

// Work buffers in HBM, sized for `capacity` ray slots (= pixels of the local image, padded to whole tiles).
struct Workspace {
    int32_t capacity;
    // per ray slot, written by the closest-hit kernel
    int32_t *slot_tri; // TriRecord index, -1 sphere, -2 miss
    float *slot_t;     // hit distance
    HitList hits[2];   // by depth & 1
    uint8_t *clear[2]; // by depth & 1: [lights*samples][capacity], 1 = the shadow ray of that sample is unoccluded
    // ray queues, rotating by depth % 3; 10 planes of `capacity` each: o.xyz, d.xyz, throughput.rgb, pixel(bits).
    // Geometry and pixel are written when the ray is spawned (compaction kernel of the parent depth), the
    // throughput later by the shading kernel of the parent depth.
    float *queue[3];
    // screen coordinates of the current pass's primary rays (prep_kernel): per image column, per local row
    float *dir_x, *dir_y;
    StageCounters *stage; // kMaxStages
    WorkCounters *work;   // [0] closest-hit walk, [1] shadow kernel, [2] screen-bin raster
    unsigned long long *shadow_shot; // kShotSlots counters, kShotStride words apart: shadow rays shot so far in this frame
    // Primary rays resolved face by face (face_raster_kernel): per pixel of the local image, the complement of (distance
    // bits, inverted rank) of the best face so far, zero = none; nullptr: not in use for this frame
    unsigned long long *pix_key;
    int32_t *tile_flag;       // face pass: per 8x8 tile of the local image, set when a face was accepted for one of its pixels
    int32_t prefilled;        // the output planes already hold what a primary miss leaves (prep_kernel): misses write nothing
    int32_t prefill_with_faces; // ... written by extra workgroups of the face pass's launch instead of by prep_kernel
    // screen binning of the primary rays; bin_entries == nullptr: not in use
    BinState *bin_state;
    int32_t *bin_count;   // per bin: faces whose projection touches it; then the fill cursor
    int32_t *bin_start;   // per bin: first entry
    int32_t *bin_entries; // TriRecord indices
    int4 *big;            // (TriRecord index, first bin x | y << 16, last bin x | y << 16, 0)
    int32_t bin_capacity; // entries
    int32_t num_bins;
    int32_t bin_max_list; // longest average list the pass accepts (kMaxAverageBinList, or no limit when bins are forced)
    // host memory the pass reports its BinState to (total, nonempty, fallback, 1), so that the host can leave the
    // binning out of the NEXT frames of a view whose lists are too long; never read by the device
    int32_t *bin_feedback;
    // ... and where every stage reports its hit and ray counts (+ 1; two ints per stage), for the sizes of later
    // frames' launches; nullptr: no report
    int32_t *stage_feedback;
    // Frames overlap on the GPU (the scene is rendered on several streams): small bounce launches keep full waves of
    // one ray per lane, which is the least work, instead of spreading the rays over more, narrower waves, which is
    // the shortest chain for a launch alone on the chip.  Same results either way.
    int32_t full_waves;
    // The binning pass of the previous frame of this view kept its bins: the tree walk behind the bins, which only
    // runs when a pass turns them down, gets a small grid (it strides over the pixels), cheap to launch and to drain.
    int32_t bins_expected;
    // Earlier frames of this view had faces in the list that the compaction tests pixel by pixel (or the scene has spheres): a hint
    // that only picks between two equivalent forms of that kernel.
    int32_t listed_faces_expected;
};

} // namespace rtamd
