// gfx950 kernels of the Flyscene render pass.
//
// One recursion level of Flyscene::traceRay (src/flyscene.cpp:581-667) is: closest hits, compaction,
// shadow rays, shading.
//
//   closest hits   intersects() = sphereIntersection + triangleIntersection (src/flyscene.cpp:669-793).
//                  Level 0 (tracePixels, src/flyscene.cpp:808-822; Camera::screenToWorld, camera.hpp:158-176):
//                  all rays leave the eye, so the faces are binned on the screen once per pass
//                  (bin_count / bin_offsets / bin_fill) and raster_kernel tests the bin of each pixel.
//                  Deeper levels, and level 0 when binning does not pay: walk_kernel, an ordered,
//                  distance-pruned walk of a four-way BVH standing in for boundingBox::intersectingBoxes
//                  (src/boundingBox.h:295-307).
//   compact_kernel gathers the rays that hit something into a dense list (hit point included) and sets
//                  up the reflection rays of traceRay (:617-646) in the queue of the next level.
//   shadow rays    getLightIntensity (src/flyscene.cpp:870-896), one any-hit ray per (hit, light sample).
//                  The direction of such a ray is the sample's position (:884), the same for every hit:
//                  shadow_grid_kernel looks the hit up in a per-sample 2D grid of the faces, built once per
//                  scene.  walk_kernel<kShadow> is the fall-back.
//   shade_kernel   shadeOffFace (:461-579) per hit, accumulation into the pixel, throughput of the
//                  reflection ray.
//
// Every candidate structure above (bins, grids, tree) is conservative and only proposes faces; a face is
// accepted by the reference's plane + three-edge arithmetic alone, and equal distances are settled by the
// faces' ranks in the reference's visiting order, so the outcome does not depend on which structure or
// which order proposed the face.
//
// Arithmetic contract: this file is compiled with -ffp-contract=off and without fast-math.  Every
// expression that decides a hit, a distance, a hit point or a shadow is written in the reference's
// evaluation order (Eigen's 3-vector reductions associate as x + (y + z)), so those values are
// bit-identical to a strict binary32 evaluation of the reference.  The slab tests, the screen projection
// and the grid projection are the only places with explicit fused multiply-adds, reciprocals or free
// evaluation order; they are conservative (intervals and boxes are widened far beyond their rounding), and
// a widened candidate test can only add faces to be tested exactly.
// Deliberate deviations, all on rays whose outcome in the reference hangs on a 0 / 0 or on the last bit of ITS box test:
// a NaN plane distance (ray inside a face plane, 0/0) is rejected, whereas the reference accepts it and then returns a
// visiting-order-dependent face; and no box ever turns a ray away from a face that the face test accepts, whereas the
// reference's rounded box test (src/boundingBox.h:257-286) may do so for a ray through a corner of a leaf's box, and its 0 / 0
// for a ray that runs in the plane of a box side turns away a box that ends there and takes one that starts there.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "device_types.h"
#include "kernels.h"

namespace rtamd {

namespace {

#ifndef RT_RASTER_PAIRS
#define RT_RASTER_PAIRS 2
#endif
#ifndef RT_KEEP_LOADS
#define RT_KEEP_LOADS 1
#endif
constexpr int kSentinel = 0x7fffffff;

enum WalkMode { kPrimary = 0, kQueue = 1, kShadow = 2 };

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
// Eigen's unrolled 3-term reduction: x + (y + z)
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// MatrixBase::normalized (Eigen/src/Core/Dot.h:124-134): divide only when the squared norm is positive
__device__ __forceinline__ V3 normalized(V3 a) {
    const float z = dot(a, a);
    if (z > 0.0f) {
        const float s = sqrtf(z);
        return mk(a.x / s, a.y / s, a.z / s);
    }
    return a;
}
__device__ __forceinline__ float norm(V3 a) { return sqrtf(dot(a, a)); }
// std::max / std::min as cwiseMax / cwiseMin apply them
__device__ __forceinline__ float smax(float a, float b) { return a < b ? b : a; }
__device__ __forceinline__ float smin(float a, float b) { return b < a ? b : a; }

// Pins a fetched value where the source puts it.  Left alone, the compiler moves a load behind the first branch that does not
// need its result (the corners of a face behind the plane test, a node's child references behind its box tests): fewer bytes
// fetched, but one more round trip to memory on the path that does need them, and these kernels wait for memory, not for
// bytes.  The empty statement claims to rewrite the registers, so everything fetched before it is fetched together.
__device__ __forceinline__ void keep(float4 &v) {
#if RT_KEEP_LOADS
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
#endif
}

__device__ __forceinline__ float *plane(float *base, int capacity, int k) { return base + (size_t)k * (size_t)capacity; }
__device__ __forceinline__ const float *plane(const float *base, int capacity, int k) { return base + (size_t)k * (size_t)capacity; }

// ---------------------------------------------------------------------------------------------
// Camera ray of one pixel sample: Camera::screenToWorld (camera.hpp:158-176, double intermediates
// kept) followed by (p - origin).normalized() (src/flyscene.cpp:822).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float screen_x(const FrameConst &fc, float px) {
    float nx = (float)(2.0 * (double)(px - fc.vp[0]) / (double)fc.vp[2] - 1.0);
    nx *= fc.ax;
    return nx;
}
__device__ __forceinline__ float screen_y(const FrameConst &fc, float py) {
    float ny = (float)(1.0 - 2.0 * (double)(py - fc.vp[1]) / (double)fc.vp[3]);
    ny *= fc.scale;
    return ny;
}

// n / d for any 32-bit n with magic = floor(2^32 / d) (0xffffffff for d == 1): the high product is the quotient or one
// below it, one correction settles it.  A division by a run-time value costs about thirty instructions otherwise.
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned d, unsigned magic) {
    unsigned q = __umulhi(n, magic);
    if (n - q * d >= d) ++q;
    return q;
}

// Image row of a local (packed) row under the row-tile partition.
__device__ __forceinline__ int global_row(const FrameConst &fc, int local_row) {
    if (fc.num_parts == 1) return local_row;
    const int tile = (int)fast_div((unsigned)local_row, (unsigned)fc.row_tile, fc.row_tile_magic);
    return (tile * fc.num_parts + fc.part) * fc.row_tile + (local_row - tile * fc.row_tile);
}

// Primary ray slots are numbered tile by tile (8x8 pixels, row-major inside the tile) so that 64
// consecutive slots are a compact screen region.  Returns false for the padding of edge tiles.
__device__ __forceinline__ bool slot_pixel(const FrameConst &fc, int slot, int &x, int &row) {
    const int tile = slot >> 6, in = slot & 63;
    const int ty = (int)fast_div((unsigned)tile, (unsigned)fc.tiles_x, fc.tiles_x_magic), tx = tile - ty * fc.tiles_x;
    x = tx * 8 + (in & 7);
    row = ty * 8 + (in >> 3);
    return x < fc.width && row < fc.local_rows;
}

// The same for a wave whose lanes hold 64 consecutive slots starting at a multiple of 64, all lanes active: the tile is
// the wave's, so its division runs once on the scalar unit instead of in every lane.
__device__ __forceinline__ bool slot_pixel_of_wave(const FrameConst &fc, int slot, int &x, int &row) {
    const int tile = __builtin_amdgcn_readfirstlane(slot >> 6), in = slot & 63;
    const int ty = (int)fast_div((unsigned)tile, (unsigned)fc.tiles_x, fc.tiles_x_magic), tx = tile - ty * fc.tiles_x;
    x = tx * 8 + (in & 7);
    row = ty * 8 + (in >> 3);
    return x < fc.width && row < fc.local_rows;
}

// Direction of the primary ray of pixel (x, local row): camera_direction with the two screen coordinates -- which
// depend on the column alone and on the row alone -- read from the tables that prep_kernel filled for this pass
// with camera_direction's own expressions (double-precision divisions once per column and row, not per pixel).
__device__ __forceinline__ V3 primary_direction_of(const FrameConst &fc, float nx, float ny) {
    const float nz = -1.0f;
    const float *m = fc.inv_view;
    const V3 w = mk(m[3] + (m[0] * nx + m[1] * ny + m[2] * nz), m[7] + (m[4] * nx + m[5] * ny + m[6] * nz),
                    m[11] + (m[8] * nx + m[9] * ny + m[10] * nz));
    return normalized(w - mk(fc.origin[0], fc.origin[1], fc.origin[2]));
}
__device__ __forceinline__ V3 primary_direction(const FrameConst &fc, const Workspace &ws, int x, int row) {
    return primary_direction_of(fc, ws.dir_x[x], ws.dir_y[row]);
}

// ---------------------------------------------------------------------------------------------
// Flyscene::sphereIntersection, src/flyscene.cpp:681-725.  pow(x, 2) promotes to double there.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool spheres_hit(const DeviceScene &sc, V3 o, V3 d, float eps, float &tbest) {
    bool any = false;
    for (int i = 0; i < sc.num_spheres; ++i) {
        const float4 s = sc.spheres[i];
        const V3 oc = mk(s.x, s.y, s.z) - o;
        const float proj = dot(oc, d);
        const float c2r = (float)((double)dot(oc, oc) - (double)proj * (double)proj);
        const float r2 = (float)((double)s.w * (double)s.w);
        if (c2r > r2) continue;
        const float h = sqrtf(r2 - c2r);
        float tin = proj - h, tout = proj + h;
        if (tin > tout) {
            const float t = tin;
            tin = tout, tout = t;
        }
        if (tin < 0 && tout < 0) continue;
        if (tin < 0) tin = tout;
        if (tin < eps || tin > tbest) continue;
        tbest = tin;
        any = true;
    }
    return any;
}

// Screen bins of this pass usable?  Not when the entry buffer overflowed, and not when the lists came out so
// long that walking the tree is cheaper (a small image of a dense mesh).  Known after bin_offsets_kernel.
__device__ __forceinline__ bool bins_accepted(const Workspace &ws) {
    const BinState *st = ws.bin_state;
    // (the big-face list is scanned by every pixel on top of its bin's list)
    return !st->fallback && (long long)st->num_big <= 4ll * (long long)ws.bin_max_list &&
           (long long)st->total <= (long long)ws.bin_max_list * (long long)max(st->nonempty, 1);
}

constexpr int kMaxHugeFaces = 192; // more faces than this in the per-pixel list: the tree walk takes the pass

// Face raster of this pass usable?  Known after face_raster_kernel.
__device__ __forceinline__ bool keys_accepted(const Workspace &ws) {
    const BinState *st = ws.bin_state;
    return !st->fallback && st->num_big <= kMaxHugeFaces;
}

__device__ __forceinline__ void flush_work(unsigned a, unsigned b, unsigned c, WorkCounters *out) {
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
        c += __shfl_xor(c, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out->box_tests, (unsigned long long)a);
        atomicAdd(&out->tri_plane, (unsigned long long)b);
        atomicAdd(&out->tri_edge, (unsigned long long)c);
    }
}

// ---------------------------------------------------------------------------------------------
// BVH walk: one wave per group of 64 rays.
//   kPrimary  ray = tile-ordered pixel slot; generated here.
//   kQueue    ray = entry of the current ray queue.
//   kShadow   group = (chunk of 64 hits, light sample): lane l shoots from hit `chunk * 64 + l` along the
//             sample POSITION (src/flyscene.cpp:884) from the un-offset hit point; any accepted face ends
//             the walk.
// The grid covers the worst case (every slot a ray, every slot a hit); waves beyond the actual count --
// which lives in device memory -- leave at once, and the hardware dispatcher balances the rest.
// Consecutive workgroups go to consecutive XCDs, so every XCD sees an interleaved sample of the whole
// image; giving each XCD a contiguous band instead was measured slower (the bands differ in cost and the
// static assignment cannot rebalance them).
// Closest-hit modes write (record index, distance) per slot; the reference's rule for equal distances
// -- the face met later in ITS visiting order wins (src/flyscene.cpp:760) -- is applied through the
// per-face rank, which makes the result independent of the order of this walk.
// Traversal stack: kLdsStack entries per lane in LDS (column per lane), deeper entries in private memory;
// the host-side builder bounds the tree depth to kStackEntries.
// ---------------------------------------------------------------------------------------------
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 8
#endif
// Waves per SIMD the walkers (and the launch the bounce walk shares with the shadow rays) are compiled for.  Eight: 64
// registers, nothing spilled; the closest-hit walk then parks six leaves per lane instead of eight, so that its 4.8 KB of
// LDS per wave fit eight times four into a CU's 160 KB (seven waves with eight parked leaves: 1 % slower with frames in
// flight, the walk alone 4 %; six waves: another 3 %).
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 8
#endif
#ifndef RT_MIN_GROUP_SHIFT
#define RT_MIN_GROUP_SHIFT 4
#endif
#ifndef RT_LEAF_QUEUE
#define RT_LEAF_QUEUE 12
#endif
#ifndef RT_LEAF_QUEUE_CLOSEST
#define RT_LEAF_QUEUE_CLOSEST 6
#endif
constexpr int kLdsStack = RT_LDS_STACK;
constexpr int kLeafQueue = RT_LEAF_QUEUE;                // parked leaves per lane, any-hit walks
constexpr int kLeafQueueClosest = RT_LEAF_QUEUE_CLOSEST; // closest-hit walks: parking delays the pruning
static_assert(kLeafQueueClosest >= 1 && kLeafQueueClosest <= kLeafQueue, "leaf queue sizes");
constexpr int kMinGroupShift = RT_MIN_GROUP_SHIFT; // narrowest group: 16 rays
#ifndef RT_SPLIT_RAYS_BELOW
#define RT_SPLIT_RAYS_BELOW (1 << 18)
#endif
#ifndef RT_SHADOW_SPLIT_BELOW
#define RT_SHADOW_SPLIT_BELOW (1 << 16)
#endif
#ifndef RT_SPLIT_LANES_SHIFT
#define RT_SPLIT_LANES_SHIFT 2
#endif
#ifndef RT_SHADOW_SPLIT_SHIFT
#define RT_SHADOW_SPLIT_SHIFT 4
#endif
constexpr int kSplitRaysBelow = RT_SPLIT_RAYS_BELOW;           // bounce launches below this many rays put several lanes on a ray
#ifndef RT_FULL_CHIP_WAVES
#define RT_FULL_CHIP_WAVES 8192
#endif
constexpr int kFullChipWaves = RT_FULL_CHIP_WAVES; // a launch with fewer groups leaves SIMDs of the 256 CUs idle

// Orders this wave's LDS traffic against its own later LDS traffic (the hardware keeps a wave's LDS operations in
// order; this keeps the compiler from moving them).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Entry `sp` of the traversal stack.  The empty asm keeps the two loads in separate blocks: merged into one
// load through a selected generic pointer they would become flat accesses.
__device__ __forceinline__ int stack_read(const int *stack, const int *deep, int sp) {
    int v;
    if (sp < kLdsStack) {
        v = stack[sp * 64];
    } else {
        v = deep[sp - kLdsStack];
        asm volatile("" : "+v"(v));
    }
    return v;
}

__device__ __forceinline__ void stack_write(int *stack, int *deep, int sp, int v) {
    if (sp < kLdsStack)
        stack[sp * 64] = v;
    else
        deep[sp - kLdsStack] = v;
}

template <int MODE, bool COUNT>
__global__ __launch_bounds__(64, RT_WAVES_PER_SIMD) void walk_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int stage, int depth, int total_static) {
    __shared__ int s_stack[(kLdsStack + kLeafQueue) * 64];
    int deep[kStackEntries - kLdsStack > 0 ? kStackEntries - kLdsStack : 1];
    constexpr bool ANY = MODE == kShadow;
    const int lane = (int)threadIdx.x;
    int *const stack = s_stack + lane;
    int *const leafq = s_stack + kLdsStack * 64 + lane;
    constexpr int CAP = MODE == kShadow ? kLeafQueue : kLeafQueueClosest;

    if (MODE == kPrimary && ws.pix_key && keys_accepted(ws)) return;         // face_raster_kernel did this pass
    if (MODE == kPrimary && ws.bin_entries && bins_accepted(ws)) return; // raster_kernel does this pass
    const int nshadow = sc.num_lights * sc.spl;
    int rays, nhits = 0;
    if (MODE == kPrimary) {
        rays = total_static;
    } else if (MODE == kQueue) {
        rays = ws.stage[stage - 1].next;
    } else {
        nhits = ws.stage[stage].hits;
        rays = nhits * nshadow;
    }
    // Lanes per group.  A launch too small to fill the chip with full waves (bounce stages) is bound by the
    // dependent-load chain of its slowest wave; narrower groups make more, shorter chains out of the same rays.
    int shift = 6;
    // (With frames in flight the other frames fill the chip, and the narrow waves and shared rays below only add
    // work: 1080p bunny, 152 K bounce rays, eight frames in flight: 0.248 -> 0.226 ms per frame with full waves,
    // while the launch alone takes 0.25 instead of 0.17 ms.  ws.full_waves says which regime the caller is in.)
    const bool alone = !ws.full_waves;
    if (MODE == kQueue && alone) // measured: no gain for shadow rays, whose groups already share origin region and direction
        while (shift > kMinGroupShift && (rays >> shift) < kFullChipWaves) --shift;
    // Bounce launches smaller still are bound by their single slowest ray (a ray skimming a surface meets a
    // hundred nodes and as many faces, one dependent fetch after the other).  Four lanes then share one ray:
    // the (up to) sixteen grandchildren of the root are dealt to them, one to start from and the rest on the
    // stack, the lanes prune against the nearest hit any of them has found so far, and their results are merged
    // under the reference's tie rule.  (Sixteen lanes shorten a lone frame's chain as well, 0.179 vs 0.166 ms for
    // 152 K rays, but cost a tenth of the throughput once frames overlap: every lane pays a wave's fixed fetches.)
    const bool split = MODE == kQueue && alone && rays < kSplitRaysBelow && fc.t_eps > 0.0f; // (shared distances are ordered by their bits)
    constexpr int kRayLanesShift = RT_SPLIT_LANES_SHIFT, kRayLanes = 1 << kRayLanesShift; // lanes that share one ray
    __shared__ int s_cut[64 >> kRayLanesShift]; // per ray of the wave: bits of the nearest accepted distance
    if (split) shift = 6 - kRayLanesShift;
    const int width = split ? 64 : 1 << shift;
    const int per_sample = MODE == kShadow ? (nhits + width - 1) >> shift : 0; // hit chunks
    const int ngroups = MODE == kShadow ? per_sample * nshadow : (rays + (1 << shift) - 1) >> shift;
    unsigned w_box = 0, w_plane = 0, w_edge = 0;
    unsigned shot = 0; // shadow rays this wave has shot

    for (int group = (int)blockIdx.x; group < ngroups; group += (int)gridDim.x) {
        const int item = split ? (group << (6 - kRayLanesShift)) + (lane >> kRayLanesShift) : (group << shift) + lane;

        V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
        bool valid = lane < width;
        int out_index;
        if (MODE == kPrimary) {
            int x, row;
            valid = slot_pixel(fc, item, x, row);
            out_index = item;
            o = mk(fc.origin[0], fc.origin[1], fc.origin[2]);
            if (valid) d = primary_direction(fc, ws, x, row);
        } else if (MODE == kQueue) {
            valid = valid && item < rays;
            out_index = item;
            if (valid) {
                float *qd = ws.queue[depth % 3];
                o = mk(plane(qd, ws.capacity, 0)[item], plane(qd, ws.capacity, 1)[item], plane(qd, ws.capacity, 2)[item]);
                d = mk(plane(qd, ws.capacity, 3)[item], plane(qd, ws.capacity, 4)[item], plane(qd, ws.capacity, 5)[item]);
            }
        } else {
            const int chunk = group / nshadow, sample = group - chunk * nshadow;
            const int hit = (chunk << shift) + lane;
            valid = valid && hit < nhits;
            out_index = sample * ws.capacity + hit;
            if (valid) {
                const float *hp = ws.hits[depth & 1].p;
                o = mk(hp[hit], hp[(size_t)ws.capacity + hit], hp[2 * (size_t)ws.capacity + hit]);
                d = mk(sc.samples[3 * sample], sc.samples[3 * sample + 1], sc.samples[3 * sample + 2]);
            }
        }
        if (MODE == kShadow) shot += (unsigned)__popcll(__ballot(valid));
        if (!valid) {
            if (MODE == kPrimary) ws.slot_tri[item] = -2; // padding of an edge tile
            continue;
        }

        // Slab test operands: t = fma(bound, inv, -origin * inv), near bound and far bound picked by the sign of
        // the direction.  The inverse is kept finite (a zero direction component becomes +-1e25) so that no
        // infinity or NaN enters the min/max chains.  origin*inv is rounded once, which puts an ABSOLUTE error of
        // up to half an ulp of |origin*inv| on every slab distance of that axis; each axis widens its own
        // interval by one ulp (folded into the addends), so a nearly axis-parallel ray, whose products are
        // astronomically large on one axis, still culls normally on the other two.
        constexpr float kInvMax = 1e25f;
        float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
        const bool cx = !(fabsf(ix) <= kInvMax), cy = !(fabsf(iy) <= kInvMax), cz = !(fabsf(iz) <= kInvMax);
        if (cx) ix = copysignf(kInvMax, d.x);
        if (cy) iy = copysignf(kInvMax, d.y);
        if (cz) iz = copysignf(kInvMax, d.z);
        // byte offsets, inside a node, of the float4 holding the ENTRY bounds of each axis (lo for a positive
        // direction, hi for a negative one); the exit bounds sit 16 bytes to the other side (offset ^ 16)
        const unsigned offx = ix < 0.0f ? 16u : 0u, offy = iy < 0.0f ? 48u : 32u, offz = iz < 0.0f ? 80u : 64u;
        const float bx = -o.x * ix, by = -o.y * iy, bz = -o.z * iz;
        // (plus the boxes' slack in space, sc.box_slack, as a distance along the ray: a ray that runs in the plane of a box side --
        // a clamped axis, both distances zero or astronomic -- passes as it does in exact arithmetic)
        const float slack = fmaxf(sc.box_slack, sc.box_slack_rel * fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)));
        const float ex = 1.2e-7f * fabsf(bx) + slack * fabsf(ix), ey = 1.2e-7f * fabsf(by) + slack * fabsf(iy);
        const float ez = 1.2e-7f * fabsf(bz) + slack * fabsf(iz);
        const float bnx = bx - ex, bny = by - ey, bnz = bz - ez; // addends of entry distances
        const float bfx = bx + ex, bfy = by + ey, bfz = bz + ez; // addends of exit distances

        float tbest = FLT_MAX;
        int best_tri = -2, best_rank = -1;
        // intersects(): spheres first, then triangles, sharing the running maximum (:673-676)
        if (sc.num_spheres > 0 && spheres_hit(sc, o, d, fc.t_eps, tbest)) best_tri = -1;
        int node = 0, sp = 0, nq = 0;
        if (ANY && best_tri == -1) node = kSentinel; // a sphere already blocks the sample
        // far limit, widened so that equal and last-bit-different distances are still visited
        float tlimit = ANY ? FLT_MAX : tbest * 1.000001f;
        int *const cut_bits = s_cut + (lane >> kRayLanesShift);
        if (split) {
            const int sub = lane & (kRayLanesShift ? kRayLanes - 1 : 0);
            node = sc.start16[sub]; // read off the tree once at scene creation: two dependent fetches less per wave
            for (int more = sub + kRayLanes; more < 16; more += kRayLanes) { // fewer lanes than start nodes: the rest waits on the stack
                const int extra = sc.start16[more];
                if (extra == kSentinel) continue;
                if (node == kSentinel) {
                    node = extra;
                } else {
                    stack_write(stack, deep, sp, extra);
                    ++sp;
                }
            }
            if (sub == 0) *cut_bits = __float_as_int(tbest);
            wave_sync(); // the other lanes of the ray read it next
        }

        for (;;) {
            float cut = FLT_MAX;
            if (split) { // nearest hit of the ray so far, whichever lane found it (distances are >= 0: bits order)
                cut = __int_as_float(*cut_bits);
                tlimit = fminf(tlimit, cut * 1.000001f);
            }
            // Node phase.  A leaf is parked in this lane's leaf queue and the descent goes on, so that the wave
            // switches between box tests and face tests as rarely as possible; a lane whose queue is full
            // waits for the others with the leaf still in `node`.
            while (node != kSentinel) {
                if (node < 0) {
                    if ((~node & 15) != 0) { // the builder's placeholder leaf holds no face: never parked
                        if (nq == CAP) break;
                        leafq[nq * 64] = node;
                        ++nq;
                    }
                    if (sp == 0) {
                        node = kSentinel;
                    } else {
                        --sp;
                        node = stack_read(stack, deep, sp);
                    }
                    continue;
                }
                // four children per node: (lo, hi) of each axis as float4 over the children, then the references.
                // Entry and exit bounds are picked by address, not by a select per value.
                const char *nodes = (const char *)sc.nodes;
                const unsigned nb = (unsigned)node << 7;
                const float4 ex4 = *(const float4 *)(nodes + (nb + offx)), fx4 = *(const float4 *)(nodes + ((nb + offx) ^ 16u));
                const float4 ey4 = *(const float4 *)(nodes + (nb + offy)), fy4 = *(const float4 *)(nodes + ((nb + offy) ^ 16u));
                const float4 ez4 = *(const float4 *)(nodes + (nb + offz)), fz4 = *(const float4 *)(nodes + ((nb + offz) ^ 16u));
                float4 chf = *(const float4 *)(nodes + (nb + 96u));
                keep(chf);
                if (COUNT) w_box += (unsigned)__float_as_int((*(const float4 *)(nodes + (nb + 112u))).x);
                const int c0 = __float_as_int(chf.x), c1 = __float_as_int(chf.y), c2 = __float_as_int(chf.z), c3 = __float_as_int(chf.w);
                float tn0, tf0, tn1, tf1, tn2, tf2, tn3, tf3;
#define RT_SLAB(K, C)                                                                                                           \
    tn##K = fmaxf(fmaxf(__builtin_fmaf(ex4.C, ix, bnx), __builtin_fmaf(ey4.C, iy, bny)), fmaxf(__builtin_fmaf(ez4.C, iz, bnz), 0.0f)); \
    tf##K = fminf(fminf(__builtin_fmaf(fx4.C, ix, bfx), __builtin_fmaf(fy4.C, iy, bfy)), fminf(__builtin_fmaf(fz4.C, iz, bfz), tlimit));
                RT_SLAB(0, x)
                RT_SLAB(1, y)
                RT_SLAB(2, z)
                RT_SLAB(3, w)
#undef RT_SLAB
                // the factor covers the relative rounding of the distances themselves
                const bool h0 = tn0 <= tf0 * 1.0000005f, h1 = tn1 <= tf1 * 1.0000005f;
                const bool h2 = tn2 <= tf2 * 1.0000005f, h3 = tn3 <= tf3 * 1.0000005f;
                const int k1 = (int)h0, k2 = k1 + (int)h1, k3 = k2 + (int)h2, hits = k3 + (int)h3;
                if (hits == 0) {
                    if (sp == 0) {
                        node = kSentinel;
                    } else {
                        --sp;
                        node = stack_read(stack, deep, sp);
                    }
                } else if (ANY) {
                    // any order will do: the first child that is hit is next, the others wait on the stack
                    if (h1 && k1 > 0) stack_write(stack, deep, sp + k1 - 1, c1);
                    if (h2 && k2 > 0) stack_write(stack, deep, sp + k2 - 1, c2);
                    if (h3 && k3 > 0) stack_write(stack, deep, sp + k3 - 1, c3);
                    node = h0 ? c0 : h1 ? c1 : h2 ? c2 : c3;
                    sp += hits - 1;
                } else {
                    // nearest first: sort (entry distance, slot) keys -- distances are >= 0, so their bit
                    // patterns order like integers; the two lowest mantissa bits carry the slot
                    int s0 = h0 ? (__float_as_int(tn0) & ~3) | 0 : 0x7fffffff;
                    int s1 = h1 ? (__float_as_int(tn1) & ~3) | 1 : 0x7fffffff;
                    int s2 = h2 ? (__float_as_int(tn2) & ~3) | 2 : 0x7fffffff;
                    int s3 = h3 ? (__float_as_int(tn3) & ~3) | 3 : 0x7fffffff;
#define RT_CSWAP(A, B)            \
    {                             \
        const int lo_ = min(A, B); \
        B = max(A, B);            \
        A = lo_;                  \
    }
                    RT_CSWAP(s0, s1)
                    RT_CSWAP(s2, s3)
                    RT_CSWAP(s0, s2)
                    RT_CSWAP(s1, s3)
                    RT_CSWAP(s1, s2)
#undef RT_CSWAP
#define RT_REF(S) (((S)&2) ? (((S)&1) ? c3 : c2) : (((S)&1) ? c1 : c0))
                    node = RT_REF(s0);
                    // farthest deepest, so that the nearest of the rest is popped first
                    if (hits > 1) stack_write(stack, deep, sp + hits - 2, RT_REF(s1));
                    if (hits > 2) stack_write(stack, deep, sp + hits - 3, RT_REF(s2));
                    if (hits > 3) stack_write(stack, deep, sp, RT_REF(s3));
#undef RT_REF
                    sp += hits - 1;
                }
            }
            if (nq == 0) break; // nothing parked: the walk of this lane is over

            // Face phase: exact plane + three-edge test of every face of every parked leaf,
            // src/flyscene.cpp:752-789
            bool blocked = false;
            int qi = 0, k = 0;
            while (qi < nq) {
                const int ref = ~leafq[qi * 64];
                const int i = (ref >> 4) + k;
                if (++k == (ref & 15)) {
                    k = 0;
                    ++qi;
                }
                float4 pn = sc.tris[i * 4 + 0];
                float4 p0 = sc.tris[i * 4 + 1];
                float4 p1 = sc.tris[i * 4 + 2];
                float4 p2 = sc.tris[i * 4 + 3];
                keep(pn), keep(p0), keep(p1), keep(p2);
                if (COUNT) ++w_plane;
                const V3 n = mk(pn.x, pn.y, pn.z);
                const float t = (pn.w - dot(o, n)) / dot(d, n);
                if (!(t >= fc.t_eps && t <= fminf(tbest, cut))) continue; // also rejects NaN (see file header)
                const int rank = __float_as_int(p1.w);
                if (!ANY && t == tbest && rank < best_rank) continue; // equal distance: the later face wins (:760)
                if (COUNT) ++w_edge;
                const V3 v0 = mk(p0.x, p0.y, p0.z), v1 = mk(p1.x, p1.y, p1.z), v2 = mk(p2.x, p2.y, p2.z);
                const V3 hp = o + t * d;
                if (dot(cross(v1 - v0, hp - v0), n) < 0) continue;
                if (dot(cross(v2 - v1, hp - v1), n) < 0) continue;
                if (dot(cross(v0 - v2, hp - v2), n) < 0) continue;
                if (ANY) {
                    blocked = true;
                    break;
                }
                tbest = t;
                best_tri = i;
                best_rank = rank;
                tlimit = t * 1.000001f;
                if (split) atomicMin(cut_bits, __float_as_int(t));
            }
            nq = 0;
            if (ANY && blocked) {
                best_tri = 0;
                break;
            }
        }

        if (split) { // merge the lanes of the ray: nearest wins, equal distances go to the later face (:760)
            for (int m = 1; m < kRayLanes; m <<= 1) {
                const float ot = __shfl_xor(tbest, m, 64);
                const int orank = __shfl_xor(best_rank, m, 64), otri = __shfl_xor(best_tri, m, 64);
                if (ot < tbest || (ot == tbest && orank > best_rank)) tbest = ot, best_rank = orank, best_tri = otri;
            }
            if ((lane & (kRayLanes - 1)) != 0) continue;
        }
        if (ANY) {
            ws.clear[depth & 1][out_index] = best_tri == -2 ? 1 : 0;
        } else {
            ws.slot_tri[out_index] = best_tri;
            ws.slot_t[out_index] = tbest;
        }
    }
    if (MODE == kShadow && lane == 0 && shot != 0) atomicAdd(ws.shadow_shot + ((int)blockIdx.x & (kShotSlots - 1)) * kShotStride, (unsigned long long)shot);
    if (COUNT) flush_work(w_box, w_plane, w_edge, ws.work + (MODE == kShadow ? 1 : 0));
}

// ---------------------------------------------------------------------------------------------
// Closest-hit walk with work sharing inside the wave.
//
// In walk_kernel a wave lasts as long as its slowest ray, and rays differ wildly: a bounce ray leaves the mesh after
// twenty-odd nodes on average, while one that skims along the surface and hits nothing pierces hundreds of leaf boxes,
// one dependent fetch after the other.  Two fifths of the lanes work on average, and a launch alone on the chip lasts
// eight times its average wave.  Here a lane that has run out of work takes a subtree from a lane that still has
// some: the donor gives away the BOTTOM entry of its traversal stack (the far child pushed nearest the root, i.e.
// the largest piece of work it holds; the bottom entries are the ones kept in LDS, where another lane can read
// them), the thief copies the donor's ray and walks that subtree for it.  All lanes working on a ray prune against
// the nearest distance any of them has accepted (s_cut, an LDS word per ray), and publish what they found under the
// reference's rule for equal distances -- the face met later in ITS visiting order wins (src/flyscene.cpp:760),
// which the per-face rank makes independent of who found what when -- through a 64-bit LDS minimum per ray:
// (distance bits, inverted rank).  The slowest ray of a wave thus gets up to 63 helpers, doubling per round.
// Narrow groups (a launch alone on the chip, see walk_kernel) start with most lanes idle, so they share from the start.
// ---------------------------------------------------------------------------------------------
#ifndef RT_STEAL_STEPS
#define RT_STEAL_STEPS 2
#endif
#ifndef RT_FACE_TRIGGER
#define RT_FACE_TRIGGER 8
#endif
struct RaySetup {
    float ix, iy, iz;
    unsigned offx, offy, offz;
    float bnx, bny, bnz, bfx, bfy, bfz;
};
// Slab-test operands of a ray; see the comment in walk_kernel.
__device__ __forceinline__ RaySetup setup_ray(V3 o, V3 d, float scene_slack, float slack_rel) {
    const float box_slack = fmaxf(scene_slack, slack_rel * fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)));
    constexpr float kInvMax = 1e25f;
    RaySetup r;
    r.ix = 1.0f / d.x, r.iy = 1.0f / d.y, r.iz = 1.0f / d.z;
    const bool cx = !(fabsf(r.ix) <= kInvMax), cy = !(fabsf(r.iy) <= kInvMax), cz = !(fabsf(r.iz) <= kInvMax);
    if (cx) r.ix = copysignf(kInvMax, d.x);
    if (cy) r.iy = copysignf(kInvMax, d.y);
    if (cz) r.iz = copysignf(kInvMax, d.z);
    r.offx = r.ix < 0.0f ? 16u : 0u, r.offy = r.iy < 0.0f ? 48u : 32u, r.offz = r.iz < 0.0f ? 80u : 64u;
    const float bx = -o.x * r.ix, by = -o.y * r.iy, bz = -o.z * r.iz;
    const float ex = 1.2e-7f * fabsf(bx) + box_slack * fabsf(r.ix), ey = 1.2e-7f * fabsf(by) + box_slack * fabsf(r.iy);
    const float ez = 1.2e-7f * fabsf(bz) + box_slack * fabsf(r.iz);
    r.bnx = bx - ex, r.bny = by - ey, r.bnz = bz - ez;
    r.bfx = bx + ex, r.bfy = by + ey, r.bfz = bz + ez;
    return r;
}
// Orders (distance, face) candidates of one ray: nearest first, equal distances by the later face.  Distances are
// positive, so their bit patterns order like unsigned integers.
__device__ __forceinline__ unsigned long long hit_key(float t, int rank) {
    return ((unsigned long long)(unsigned)__float_as_int(t) << 32) | (unsigned long long)(unsigned)(0x7fffffff - rank);
}

// LDS of one wave of the closest-hit walk (passed in by the kernel, so that a kernel whose workgroups take different roles can
// lay the roles' LDS over each other).
struct WalkLds {
    unsigned long long key[64]; // per ray of the wave: best (distance, rank) published so far
    int stack[(kLdsStack + kLeafQueueClosest) * 64];
    int tri[64];                // ... and its record
    int cut[64];                // ... and the bits of the nearest distance accepted by any lane, published or not
    int donor[64];
};

// (`block` of `nblocks`: the launch's block index and grid, or this kernel's share of a launch it shares with another)
template <int MODE, bool COUNT>
__device__ __forceinline__ void walk_steal_body(const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, int total_static,
                                                int block, int nblocks, WalkLds &lds) {
    static_assert(MODE == kPrimary || MODE == kQueue, "closest-hit modes only");
    int *const s_stack = lds.stack;
    unsigned long long *const s_key = lds.key;
    int *const s_tri = lds.tri, *const s_cut = lds.cut, *const s_donor = lds.donor;
    int deep[kStackEntries - kLdsStack > 0 ? kStackEntries - kLdsStack : 1];
    const int lane = (int)threadIdx.x;
    int *const stack = s_stack + lane;
    int *const leafq = s_stack + kLdsStack * 64 + lane;
    constexpr int CAP = kLeafQueueClosest;
    const unsigned long long below = (1ull << lane) - 1ull;

    if (MODE == kPrimary && ws.pix_key && keys_accepted(ws)) return;         // face_raster_kernel did this pass
    if (MODE == kPrimary && ws.bin_entries && bins_accepted(ws)) return; // raster_kernel does this pass
    const int rays = MODE == kPrimary ? total_static : ws.stage[stage - 1].next;
    int shift = 6;
    if (MODE == kQueue && !ws.full_waves)
        while (shift > kMinGroupShift && (rays >> shift) < kFullChipWaves) --shift;
    const int ngroups = (rays + (1 << shift) - 1) >> shift;
    unsigned w_box = 0, w_plane = 0, w_edge = 0;

    for (int group = block; group < ngroups; group += nblocks) {
        const int item = (group << shift) + lane;
        V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
        bool valid = lane < (1 << shift);
        if (MODE == kPrimary) {
            int x, row;
            valid = slot_pixel(fc, item, x, row);
            o = mk(fc.origin[0], fc.origin[1], fc.origin[2]);
            if (valid)
                d = primary_direction(fc, ws, x, row);
            else
                ws.slot_tri[item] = -2; // padding of an edge tile
        } else {
            valid = valid && item < rays;
            if (valid) {
                const float *qd = ws.queue[depth % 3];
                o = mk(plane(qd, ws.capacity, 0)[item], plane(qd, ws.capacity, 1)[item], plane(qd, ws.capacity, 2)[item]);
                d = mk(plane(qd, ws.capacity, 3)[item], plane(qd, ws.capacity, 4)[item], plane(qd, ws.capacity, 5)[item]);
            }
        }
        float tbest = FLT_MAX;
        int best_tri = -2, best_rank = -1;
        // intersects(): spheres first, then triangles, sharing the running maximum (:673-676)
        if (valid && sc.num_spheres > 0 && spheres_hit(sc, o, d, fc.t_eps, tbest)) best_tri = -1;
        s_key[lane] = best_tri == -1 ? hit_key(tbest, -1) : ~0ull;
        s_tri[lane] = best_tri;
        s_cut[lane] = __float_as_int(tbest);
        int owner = lane; // the ray this lane works for = the lane that loaded it
        RaySetup rs = setup_ray(o, d, sc.box_slack, sc.box_slack_rel);
        int node = valid ? 0 : kSentinel, sp = 0, base = 0, nq = 0;
        bool dirty = false; // holds an accepted face that is not published yet
        float tlimit = tbest * 1.000001f;
        wave_sync();

        for (;;) {
            const bool has_work = node != kSentinel || sp > base || nq > 0;
            // ---- lanes out of work publish what they found
            if (__ballot(!has_work && dirty) != 0ull) {
                const unsigned long long key = hit_key(tbest, best_rank);
                if (!has_work && dirty) atomicMin(&s_key[owner], key);
                wave_sync();
                if (!has_work && dirty) {
                    if (s_key[owner] == key) s_tri[owner] = best_tri;
                    dirty = false;
                }
                wave_sync();
            }
            const unsigned long long work = __ballot(has_work);
            if (work == 0ull) break;
            // ---- ... and take the bottom of a busy lane's stack
            const bool can_give = node != kSentinel && sp > base && base < kLdsStack;
            const unsigned long long donors = __ballot(can_give), idle = ~work;
            if (idle != 0ull && donors != 0ull) {
                const int n = min((int)__popcll(idle), (int)__popcll(donors));
                const int rank_d = (int)__popcll(donors & below), rank_i = (int)__popcll(idle & below);
                const bool gives = can_give && rank_d < n, thief = !has_work && rank_i < n;
                if (gives) s_donor[rank_d] = lane;
                wave_sync();
                const int src = thief ? s_donor[rank_i] : lane;
                const float sox = __shfl(o.x, src, 64), soy = __shfl(o.y, src, 64), soz = __shfl(o.z, src, 64);
                const float sdx = __shfl(d.x, src, 64), sdy = __shfl(d.y, src, 64), sdz = __shfl(d.z, src, 64);
                const int sowner = __shfl(owner, src, 64), sbase = __shfl(base, src, 64);
                if (thief) {
                    o = mk(sox, soy, soz), d = mk(sdx, sdy, sdz);
                    owner = sowner;
                    rs = setup_ray(o, d, sc.box_slack, sc.box_slack_rel);
                    node = s_stack[sbase * 64 + src];
                    sp = base = 0;
                    tbest = __int_as_float(s_cut[owner]); // nothing farther is of interest; equal distances still are (rank)
                    best_tri = -2, best_rank = -1;
                    tlimit = tbest * 1.000001f;
                }
                if (gives) ++base;
            }

            // ---- node phase, a few steps at a time so that idle lanes get their chance
            {
                const float cut = __int_as_float(s_cut[owner]);
                tlimit = fminf(tlimit, cut * 1.000001f);
            }
            for (int steps = 0; node != kSentinel && steps < RT_STEAL_STEPS; ++steps) {
                if (node < 0) {
                    if ((~node & 15) != 0) { // the builder's placeholder leaf holds no face: never parked
                        if (nq == CAP) break;
                        leafq[nq * 64] = node;
                        ++nq;
                    }
                    if (sp == base) {
                        node = kSentinel;
                    } else {
                        --sp;
                        node = stack_read(stack, deep, sp);
                    }
                    continue;
                }
                const char *nodes = (const char *)sc.nodes;
                const unsigned nb = (unsigned)node << 7;
                const float4 ex4 = *(const float4 *)(nodes + (nb + rs.offx)), fx4 = *(const float4 *)(nodes + ((nb + rs.offx) ^ 16u));
                const float4 ey4 = *(const float4 *)(nodes + (nb + rs.offy)), fy4 = *(const float4 *)(nodes + ((nb + rs.offy) ^ 16u));
                const float4 ez4 = *(const float4 *)(nodes + (nb + rs.offz)), fz4 = *(const float4 *)(nodes + ((nb + rs.offz) ^ 16u));
                float4 chf = *(const float4 *)(nodes + (nb + 96u));
                keep(chf);
                if (COUNT) w_box += (unsigned)__float_as_int((*(const float4 *)(nodes + (nb + 112u))).x);
                const int c0 = __float_as_int(chf.x), c1 = __float_as_int(chf.y), c2 = __float_as_int(chf.z), c3 = __float_as_int(chf.w);
                float tn0, tf0, tn1, tf1, tn2, tf2, tn3, tf3;
#define RT_SLAB(K, C)                                                                                                                          \
    tn##K = fmaxf(fmaxf(__builtin_fmaf(ex4.C, rs.ix, rs.bnx), __builtin_fmaf(ey4.C, rs.iy, rs.bny)), fmaxf(__builtin_fmaf(ez4.C, rs.iz, rs.bnz), 0.0f)); \
    tf##K = fminf(fminf(__builtin_fmaf(fx4.C, rs.ix, rs.bfx), __builtin_fmaf(fy4.C, rs.iy, rs.bfy)), fminf(__builtin_fmaf(fz4.C, rs.iz, rs.bfz), tlimit));
                RT_SLAB(0, x)
                RT_SLAB(1, y)
                RT_SLAB(2, z)
                RT_SLAB(3, w)
#undef RT_SLAB
                const bool h0 = tn0 <= tf0 * 1.0000005f, h1 = tn1 <= tf1 * 1.0000005f;
                const bool h2 = tn2 <= tf2 * 1.0000005f, h3 = tn3 <= tf3 * 1.0000005f;
                const int hits = (int)h0 + (int)h1 + (int)h2 + (int)h3;
                if (hits == 0) {
                    if (sp == base) {
                        node = kSentinel;
                    } else {
                        --sp;
                        node = stack_read(stack, deep, sp);
                    }
                } else {
                    int s0 = h0 ? (__float_as_int(tn0) & ~3) | 0 : 0x7fffffff;
                    int s1 = h1 ? (__float_as_int(tn1) & ~3) | 1 : 0x7fffffff;
                    int s2 = h2 ? (__float_as_int(tn2) & ~3) | 2 : 0x7fffffff;
                    int s3 = h3 ? (__float_as_int(tn3) & ~3) | 3 : 0x7fffffff;
#define RT_CSWAP(A, B)             \
    {                              \
        const int lo_ = min(A, B); \
        B = max(A, B);             \
        A = lo_;                   \
    }
                    RT_CSWAP(s0, s1)
                    RT_CSWAP(s2, s3)
                    RT_CSWAP(s0, s2)
                    RT_CSWAP(s1, s3)
                    RT_CSWAP(s1, s2)
#undef RT_CSWAP
#define RT_REF(S) (((S)&2) ? (((S)&1) ? c3 : c2) : (((S)&1) ? c1 : c0))
                    node = RT_REF(s0);
                    if (hits > 1) stack_write(stack, deep, sp + hits - 2, RT_REF(s1));
                    if (hits > 2) stack_write(stack, deep, sp + hits - 3, RT_REF(s2));
                    if (hits > 3) stack_write(stack, deep, sp, RT_REF(s3));
#undef RT_REF
                    sp += hits - 1;
                }
            }

            // ---- face phase: once no lane can descend any further, or enough of them wait for it; every lane with
            // parked leaves joins.  Exact plane + three-edge test, src/flyscene.cpp:752-789.
            const bool descending = node != kSentinel && !(node < 0 && nq == CAP);
            const bool waiting = nq > 0 && !descending;
            const unsigned long long m_desc = __ballot(descending), m_wait = __ballot(waiting);
            if (m_wait != 0ull && (m_desc == 0ull || (int)__popcll(m_wait) >= RT_FACE_TRIGGER)) {
                const float cut = __int_as_float(s_cut[owner]);
                int qi = 0, k = 0;
                while (qi < nq) {
                    const int ref = ~leafq[qi * 64];
                    const int i = (ref >> 4) + k;
                    if (++k == (ref & 15)) {
                        k = 0;
                        ++qi;
                    }
                    float4 pn = sc.tris[i * 4 + 0];
                    float4 p0 = sc.tris[i * 4 + 1];
                    float4 p1 = sc.tris[i * 4 + 2];
                    float4 p2 = sc.tris[i * 4 + 3];
                    keep(pn), keep(p0), keep(p1), keep(p2);
                    if (COUNT) ++w_plane;
                    const V3 n = mk(pn.x, pn.y, pn.z);
                    const float t = (pn.w - dot(o, n)) / dot(d, n);
                    if (!(t >= fc.t_eps && t <= fminf(tbest, cut))) continue; // also rejects NaN (see file header)
                    const int rank = __float_as_int(p1.w);
                    if (t == tbest && rank < best_rank) continue; // equal distance: the later face wins (:760)
                    if (COUNT) ++w_edge;
                    const V3 v0 = mk(p0.x, p0.y, p0.z), v1 = mk(p1.x, p1.y, p1.z), v2 = mk(p2.x, p2.y, p2.z);
                    const V3 hp = o + t * d;
                    if (dot(cross(v1 - v0, hp - v0), n) < 0) continue;
                    if (dot(cross(v2 - v1, hp - v1), n) < 0) continue;
                    if (dot(cross(v0 - v2, hp - v2), n) < 0) continue;
                    tbest = t;
                    best_tri = i;
                    best_rank = rank;
                    tlimit = t * 1.000001f;
                    dirty = true;
                    atomicMin(&s_cut[owner], __float_as_int(t));
                }
                nq = 0;
            }
        }

        wave_sync();
        if (valid) {
            const unsigned long long key = s_key[lane];
            ws.slot_tri[item] = s_tri[lane];
            ws.slot_t[item] = key == ~0ull ? FLT_MAX : __int_as_float((int)(unsigned)(key >> 32));
        }
        wave_sync();
    }
    if (COUNT) flush_work(w_box, w_plane, w_edge, ws.work + 0);
}

template <int MODE, bool COUNT>
__global__ __launch_bounds__(64, RT_WAVES_PER_SIMD) void walk_steal_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int stage, int depth, int total_static) {
    __shared__ WalkLds lds;
    walk_steal_body<MODE, COUNT>(sc, fc, ws, stage, depth, total_static, (int)blockIdx.x, (int)gridDim.x, lds);
}

// ---------------------------------------------------------------------------------------------
// Primary rays by screen binning.  All primary rays leave one point, so the faces a pixel's ray can hit are
// those whose perspective projection covers the pixel.  Per pass: every face is projected, its pixel bounding
// box (widened far beyond the rounding of the projection and of the exact test) is entered into the bins
// of kBinPixels^2 pixels it touches (count, claim ranges, fill); then each pixel runs the exact plane +
// three-edge test over the list of its bin and keeps the nearest face under the reference's tie rule.  The
// exact test decides, so the result is the one the tree walk gives.  Faces with a vertex at or behind the
// eye plane, or covering more than kBigBins bins, go to a short list that every pixel scans with a box
// check.  When the entry buffer would overflow (BinState::fallback), or the lists come out longer on average than
// a tree walk is worth (bins_accepted), the pass leaves its bins alone and the tree walk runs.
// ---------------------------------------------------------------------------------------------
struct PixelBox {
    int x0, y0, x1, y1; // bins, inclusive, clamped to the image
    bool empty, big;
    // The projected triangle, for faces whose box spans several bins: a bin none of whose pixels can lie inside it
    // (face_touches_bin) is left out.  Edge k runs from (ex[k], ey[k]) along (dx[k], dy[k]), oriented so that the
    // inside is on its left; coordinates are pixel indices (the sample offset of the pass is taken off).
    bool cull;
    float margin;
    float ex[3], ey[3], dx[3], dy[3];
};

// Widening of the projected box and triangle, in pixels (FrameConst::raster_margin, set by the host to
// max(kBinMarginFloor, four times the bound below)).
//
// Bound on the error of a projected corner, u = 2^-24 the unit roundoff.  project_point forms r = p - o, q = M r with M the
// float rounding of the inverse camera matrix, s = q.z * (1 / (e.z - 1)), n = q.xy * rcp(s) - e.xy and the pixel as an
// offset plus n times a scale; the reciprocals and scales are rounded once on the host (u each), the hardware reciprocal is
// good to one unit in the last place (2 u), every product and sum adds u.  (Only the candidates depend on these figures;
// whether a pixel is hit is decided by the reference's own arithmetic in the exact test.)
//   each component of q is off by at most 4 u R, R = |q.x| + |q.y| + |q.z| (three products, two sums, the rounding of M);
//   s inherits that plus two roundings: 4 u R + 2 u s;
//   n.x = q.x rcp(s) - e.x is then off by at most (8 R / s + 6) u (1 + |n.x|) / 2 <= (8 R / s + 6) u for a corner on the screen,
//   where |n.x|, |n.y| <= 1 (tangent of half the field of view times the aspect) and so R <= 3 s;
//   the pixel is n scaled by (W / 2) / (aspect tan(fovy / 2)) = (H / 2) / tan(fovy / 2) = 0.87 H at the reference's 60 degrees,
//   with three more roundings of a number of at most W.
// So a corner ON the screen lands within 0.87 H * 30 u + 3 u W of its true pixel position; the host takes
// 0.87 H * 32 u + 6 u W: 1.7e-6 H + 3.6e-7 W, i.e. 0.0025 pixel at 1080p and 0.010 pixel at 8K.  A corner far OFF the screen
// may be off by much more in the coordinate that
// is on the screen (R / s is large), but only the box's extremes use it, and a wrong extreme at a corner changes which
// pixels are tested only next to that corner, which is off the screen.  Corners with s <= 1e-6 R count as unprojectable
// and send the face to the per-pixel list; the edge functions are only used with every corner within 16384 pixels
// (R / s <= 20 at 1080p: 0.009 pixel).  The exact test's own rounding moves a face's boundary by a few u of the face's
// size in pixels (0.001 pixel for a face 10^4 pixels wide).  The floor of a sixteenth of a pixel is six times the 8K bound;
// beyond that the host scales the margin with the image.
// All of this takes coordinates of the size of the depths.  The exact test computes with the coordinates themselves (the hit
// point is eye + t * direction, rounded to the last place of the EYE's coordinates; the plane's D and eye . n cancel at that
// size), so where the eye and the faces lie far from the origin and near each other its decisions are only good to about four
// units in the last place of the largest coordinate: 2.4e-4 for an eye a thousand units out, a quarter of a pixel for a face two
// units away at 4K.  A face's margin is therefore raster_margin + coord_slack * spread, coord_slack = 2^-19 (sixteen such
// units) * largest coordinate * pixels per unit of n, spread = (1 + |n.x| + |n.y|) / s of the face's nearest corner: a thousandth
// of a pixel for the reference's own scenes (coordinates within the unit cube, depths of one or two), tens of pixels ten
// thousand units out, where the faces' boxes outgrow the pass and it gives way to the tree.
// It matters that the margin is not larger than needed for faces smaller than a pixel (a distant mesh): their box holds a
// pixel centre with probability (size + 2 margin)^2, and a quarter of a pixel made the work 3-4 times what it had to be.
#ifndef RT_BIN_MARGIN
#define RT_BIN_MARGIN 0.0625f
#endif
constexpr float kBinMarginFloor = RT_BIN_MARGIN;

// pixel position of a world point: inverse of camera_direction.  false: the point is not in front of the eye.
// `behind`: the point lies behind the eye plane by far more than any rounding here or in the exact test.
// `spread`: pixels that a displacement of the point by one unit of length moves its image by, at most (sideways 1 / s of the
// projection's scale, in depth |n| / s of it), in units of the projection's scale.
__device__ __forceinline__ bool project_point(const FrameConst &fc, V3 p, float &px, float &py, bool &behind, float &spread) {
    spread = 0.0f;
    const V3 r = p - mk(fc.origin[0], fc.origin[1], fc.origin[2]);
    const float *m = fc.lin_inv;
    const float qx = m[0] * r.x + (m[1] * r.y + m[2] * r.z);
    const float qy = m[3] * r.x + (m[4] * r.y + m[5] * r.z);
    const float qz = m[6] * r.x + (m[7] * r.y + m[8] * r.z);
    // r = s * L * (n + e) with n.z = -1  ->  q = s * (n + e)
    const float s = qz * fc.proj_s;
    const float reach = fabsf(qx) + fabsf(qy) + fabsf(qz);
    behind = s < -1e-4f * reach;
    if (!(s > 1e-6f * reach)) return false;
    const float rs = __builtin_amdgcn_rcpf(s); // (no division on this path needs to be correctly rounded: see the bound above)
    const float nx = qx * rs - fc.cam_e[0], ny = qy * rs - fc.cam_e[1];
    px = fc.px_off + nx * fc.px_scale;
    py = fc.py_off - ny * fc.py_scale;
    spread = fabsf(rs) * (1.0f + fabsf(qx * rs) + fabsf(qy * rs));
    return px == px && py == py;
}

// Projection of one face: the integer pixels [px0, px1] x [py0, py1] (clamped to the image) whose sample points can
// lie inside the projected triangle widened by the margin, and the projected corners themselves.
struct ProjectedFace {
    bool empty;     // no pixel of the image can see the face
    bool unbounded; // a corner beside or behind the eye: the box is the whole image
    float x0, y0, x1, y1, x2, y2;
    int px0, py0, px1, py1;
    float margin; // widening of the face's box and projected triangle, in pixels: the pass's margin plus the term of the coordinates' size
};
__device__ __forceinline__ ProjectedFace project_face(const DeviceScene &sc, const FrameConst &fc, int tri) {
    ProjectedFace f;
    f.empty = false, f.unbounded = false;
    f.x0 = f.y0 = f.x1 = f.y1 = f.x2 = f.y2 = 0.0f;
    f.px0 = f.py0 = 0, f.px1 = f.py1 = -1;
    f.margin = fc.raster_margin;
    const float4 p0 = sc.tris[tri * 4 + 1], p1 = sc.tris[tri * 4 + 2], p2 = sc.tris[tri * 4 + 3];
    bool back0, back1, back2;
    float spread0, spread1, spread2;
    const bool ok0 = project_point(fc, mk(p0.x, p0.y, p0.z), f.x0, f.y0, back0, spread0);
    const bool ok1 = project_point(fc, mk(p1.x, p1.y, p1.z), f.x1, f.y1, back1, spread1);
    const bool ok2 = project_point(fc, mk(p2.x, p2.y, p2.z), f.x2, f.y2, back2, spread2);
    // Every point a primary ray reaches at a positive distance lies in front of the eye plane (s > 0 above, and s is
    // linear in the point); a face with all three corners clearly behind that plane lies behind it entirely and can
    // meet no such ray.  Without this, a view from inside or past the mesh sent those faces to the list every pixel scans.
    if (back0 && back1 && back2) {
        f.empty = true;
        return f;
    }
    float xlo = 0.0f, ylo = 0.0f, xhi = (float)fc.width, yhi = (float)fc.height;
    if (ok0 && ok1 && ok2) {
        // pixel (x, y) samples at (x + sub_x, y + sub_y): shift the box instead of the samples; the margin covers
        // the rounding of the projection (about 1e-3 pixel at 8K) and of the exact test many times over
        // (fc.coord_slack: what the exact test's rounding of coordinates of the scene's size is in pixels at unit spread)
        const float margin = f.margin = fc.raster_margin + fc.coord_slack * fmaxf(fmaxf(spread0, spread1), spread2);
        xlo = fminf(fminf(f.x0, f.x1), f.x2) - margin - fc.sub_x, xhi = fmaxf(fmaxf(f.x0, f.x1), f.x2) + margin - fc.sub_x;
        ylo = fminf(fminf(f.y0, f.y1), f.y2) - margin - fc.sub_y, yhi = fmaxf(fmaxf(f.y0, f.y1), f.y2) + margin - fc.sub_y;
    } else {
        f.unbounded = true;
    }
    // integer pixels inside [lo, hi]
    const float fx0 = fmaxf(ceilf(xlo), 0.0f), fy0 = fmaxf(ceilf(ylo), 0.0f);
    const float fx1 = fminf(floorf(xhi), (float)(fc.width - 1)), fy1 = fminf(floorf(yhi), (float)(fc.height - 1));
    if (!(fx0 <= fx1 && fy0 <= fy1)) {
        f.empty = true;
        return f;
    }
    f.px0 = (int)fx0, f.py0 = (int)fy0, f.px1 = (int)fx1, f.py1 = (int)fy1;
    return f;
}

__device__ __forceinline__ PixelBox face_bins(const DeviceScene &sc, const FrameConst &fc, int tri) {
    PixelBox b;
    b.empty = false, b.big = false, b.cull = false;
    const ProjectedFace pf = project_face(sc, fc, tri);
    b.margin = pf.margin;
    if (pf.empty) {
        b.empty = true;
        b.x0 = b.y0 = 0, b.x1 = b.y1 = -1;
        return b;
    }
    const bool ok = !pf.unbounded;
    if (!ok) b.big = true; // a vertex beside or behind the eye: its projection is unbounded
    const float x0 = pf.x0, y0 = pf.y0, x1 = pf.x1, y1 = pf.y1, x2 = pf.x2, y2 = pf.y2;
    const int fx0 = pf.px0, fy0 = pf.py0, fx1 = pf.px1, fy1 = pf.py1;
    b.x0 = fx0 / kBinPixels, b.x1 = fx1 / kBinPixels, b.y0 = fy0 / kBinPixels, b.y1 = fy1 / kBinPixels;
    if ((b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1) > kBigBins) b.big = true;
    // The box of a triangle holds about twice the bins the triangle touches.  Worth three edge evaluations per bin
    // once the box spans 2 x 2 bins; only with all vertices near the screen (the edge functions are products of pixel
    // coordinates: their rounding must stay far below the margin term) and a triangle that has an inside.
    b.cull = false;
    if (ok && !b.big && b.x1 > b.x0 && b.y1 > b.y0) {
        const float lim = 16384.0f;
        const float area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
        const bool near_screen = fabsf(x0) < lim && fabsf(y0) < lim && fabsf(x1) < lim && fabsf(y1) < lim && fabsf(x2) < lim && fabsf(y2) < lim;
        if (near_screen && fabsf(area) > 1e-3f) {
            b.cull = true;
            const float px[3] = {x0 - fc.sub_x, x1 - fc.sub_x, x2 - fc.sub_x}, py[3] = {y0 - fc.sub_y, y1 - fc.sub_y, y2 - fc.sub_y};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int n = k == 2 ? 0 : k + 1;
                // counter-clockwise (area > 0): inside is left of p_k -> p_n; clockwise: left of p_n -> p_k
                const int from = area > 0.0f ? k : n, to = area > 0.0f ? n : k;
                b.ex[k] = px[from], b.ey[k] = py[from];
                b.dx[k] = px[to] - px[from], b.dy[k] = py[to] - py[from];
            }
        }
    }
    return b;
}

// Can a pixel of bin (x, y) lie inside the projected triangle widened by the margin?  E(r) = dx (r.y - ey) - dy (r.x - ex)
// is >= 0 left of an edge; moving r by up to the margin in x and y changes E by at most margin (|dx| + |dy|), so a bin
// whose most favourable corner stays below minus that lies outside for good.
__device__ __forceinline__ bool face_touches_bin(const PixelBox &b, int x, int y) {
    if (!b.cull) return true;
    const float rx0 = (float)(x * kBinPixels), ry0 = (float)(y * kBinPixels);
    const float rx1 = rx0 + (float)(kBinPixels - 1), ry1 = ry0 + (float)(kBinPixels - 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float cy = b.dx[k] >= 0.0f ? ry1 : ry0; // E grows with r.y when dx > 0 ...
        const float cx = b.dy[k] >= 0.0f ? rx0 : rx1; // ... and falls with r.x when dy > 0
        const float e = b.dx[k] * (cy - b.ey[k]) - b.dy[k] * (cx - b.ex[k]);
        if (e < -b.margin * (fabsf(b.dx[k]) + fabsf(b.dy[k]))) return false;
    }
    return true;
}

// does this rank render any row of bin row `by`?  (row tiles are dealt round-robin, see global_row)
__device__ __forceinline__ bool bin_row_owned(const FrameConst &fc, int by) {
    if (fc.num_parts == 1) return true;
    const int r0 = by * kBinPixels, r1 = min(r0 + kBinPixels - 1, fc.height - 1);
    for (int t = r0 / fc.row_tile; t <= r1 / fc.row_tile; ++t)
        if (t % fc.num_parts == fc.part) return true;
    return false;
}

// atomicAdd(base + bin, 1) for every active lane, lanes of the wave that name the same bin sharing ONE atomic
// (neighbouring faces land in the same bins, and atomics that cross the XCDs are what the binning passes pay for).
// Returns the value the lane's own atomicAdd would have returned.  Every lane of the wave must call this.
// The groups are found first (ballots only); then the leaders of ALL groups issue their atomics in one instruction,
// so a step costs one round trip to the atomic unit, not one per distinct bin.
__device__ __forceinline__ int shared_increment(int *base, int bin, bool active) {
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long todo = __ballot(active);
    int leader_of = lane, rank = 0, size = 0;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int b = __shfl(bin, leader, 64);
        const bool mine = active && bin == b;
        const unsigned long long same = __ballot(mine);
        if (mine) leader_of = leader, rank = (int)__popcll(same & below), size = (int)__popcll(same);
        todo &= ~same;
    }
    int old = 0;
    if (active && leader_of == lane) old = atomicAdd(base + bin, size);
    old = __shfl(old, leader_of, 64);
    return old + rank;
}

// AGG: the wave steps through its faces' bins together and shares the atomics.  It pays when the faces are small
// and many (neighbouring faces name the same bins; 1.1 M faces at 4K: 7 % more frames per second), and costs when
// they are few and large (every lane waits for the face with the most bins).
template <bool AGG>
__global__ __launch_bounds__(256) void bin_count_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int num_tris) {
    const int tri = (int)(blockIdx.x * 256 + threadIdx.x);
    if (AGG) {
    PixelBox b;
    b.empty = true, b.big = false, b.cull = false, b.x0 = b.y0 = 0, b.x1 = b.y1 = -1;
    if (tri < num_tris) b = face_bins(sc, fc, tri);
    if (!b.empty && b.big) {
        const int at = atomicAdd(&ws.bin_state->num_big, 1);
        ws.big[at] = make_int4(tri, b.x0 | (b.y0 << 16), b.x1 | (b.y1 << 16), 0);
    }
    const int w = b.x1 - b.x0 + 1, count = (b.empty || b.big) ? 0 : w * (b.y1 - b.y0 + 1);
    for (int j = 0; __ballot(j < count) != 0; ++j) { // the wave steps through its faces' bins together
        const bool active = j < count;
        const int y = b.y0 + (active ? j / w : 0), x = b.x0 + (active ? j % w : 0);
        const bool owned = active && bin_row_owned(fc, y) && face_touches_bin(b, x, y);
        (void)shared_increment(ws.bin_count, y * fc.bins_x + x, owned);
    }
    return;
    }
    if (tri >= num_tris) return;
    const PixelBox b = face_bins(sc, fc, tri);
    if (b.empty) return;
    if (b.big) {
        const int at = atomicAdd(&ws.bin_state->num_big, 1);
        ws.big[at] = make_int4(tri, b.x0 | (b.y0 << 16), b.x1 | (b.y1 << 16), 0);
        return;
    }
    for (int y = b.y0; y <= b.y1; ++y) {
        if (!bin_row_owned(fc, y)) continue;
        for (int x = b.x0; x <= b.x1; ++x)
            if (face_touches_bin(b, x, y)) atomicAdd(&ws.bin_count[y * fc.bins_x + x], 1);
    }
}

// Claims a contiguous range of the entry buffer for every bin: prefix sum inside a workgroup, one atomic per
// workgroup.  A workgroup takes 2048 bins, eight per thread: the atomics all name one address and serialise in L2
// (2025 of them, one per 256 bins of a 4K image, took 24 us; 8K has four times the bins).  Leaves bin_count = 0
// for the fill pass to count up again.
constexpr int kOffsetItems = 8;
__global__ __launch_bounds__(256) void bin_offsets_kernel(Workspace ws) {
    __shared__ int s_wave[4], s_full[4];
    __shared__ int s_base;
    const int bin0 = (int)(blockIdx.x * 256 + threadIdx.x) * kOffsetItems;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    int c[kOffsetItems];
    int mine = 0, full = 0;
#pragma unroll
    for (int k = 0; k < kOffsetItems; ++k) {
        c[k] = bin0 + k < ws.num_bins ? ws.bin_count[bin0 + k] : 0;
        mine += c[k];
        full += c[k] > 0 ? 1 : 0;
    }
    // inclusive scan of the threads' sums over the wave, and the wave's count of non-empty bins
    int incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    for (int d = 32; d > 0; d >>= 1) full += __shfl_xor(full, d, 64);
    if (lane == 63) s_wave[wave] = incl;
    if (lane == 0) s_full[wave] = full;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long n = (long long)s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        unsigned long long base = n ? atomicAdd(&ws.bin_state->total, (unsigned long long)n) : 0ull;
        if (s_full[0] + s_full[1] + s_full[2] + s_full[3]) atomicAdd(&ws.bin_state->nonempty, s_full[0] + s_full[1] + s_full[2] + s_full[3]);
        if (base + (unsigned long long)n > (unsigned long long)ws.bin_capacity) { // (the fill pass then leaves the bins alone)
            ws.bin_state->fallback = 1;
            base = 0;
        }
        s_base = (int)base;
    }
    __syncthreads();
    int start = s_base + incl - mine;
    for (int w = 0; w < wave; ++w) start += s_wave[w];
#pragma unroll
    for (int k = 0; k < kOffsetItems; ++k) {
        if (bin0 + k < ws.num_bins) {
            ws.bin_start[bin0 + k] = start;
            ws.bin_count[bin0 + k] = 0;
        }
        start += c[k];
    }
}

template <bool AGG>
__global__ __launch_bounds__(256) void bin_fill_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int num_tris) {
    const int tri = (int)(blockIdx.x * 256 + threadIdx.x);
    if (tri == 0 && ws.bin_feedback) {
        ws.bin_feedback[0] = (int)min(ws.bin_state->total, 0x7fffffffull), ws.bin_feedback[1] = ws.bin_state->nonempty;
        ws.bin_feedback[2] = bins_accepted(ws) ? 0 : 1, ws.bin_feedback[3] = 1;
    }
    if (!bins_accepted(ws)) return;
    if (AGG) {
    PixelBox b;
    b.empty = true, b.big = false, b.cull = false, b.x0 = b.y0 = 0, b.x1 = b.y1 = -1;
    if (tri < num_tris) b = face_bins(sc, fc, tri);
    const int w = b.x1 - b.x0 + 1, count = (b.empty || b.big) ? 0 : w * (b.y1 - b.y0 + 1);
    for (int j = 0; __ballot(j < count) != 0; ++j) {
        const bool active = j < count;
        const int y = b.y0 + (active ? j / w : 0), x = b.x0 + (active ? j % w : 0);
        const bool owned = active && bin_row_owned(fc, y) && face_touches_bin(b, x, y);
        const int bin = y * fc.bins_x + x;
        const int at = shared_increment(ws.bin_count, bin, owned);
        if (owned) ws.bin_entries[ws.bin_start[bin] + at] = tri;
    }
    return;
    }
    if (tri >= num_tris) return;
    const PixelBox b = face_bins(sc, fc, tri);
    if (b.empty || b.big) return;
    for (int y = b.y0; y <= b.y1; ++y) {
        if (!bin_row_owned(fc, y)) continue;
        for (int x = b.x0; x <= b.x1; ++x) {
            if (!face_touches_bin(b, x, y)) continue;
            const int bin = y * fc.bins_x + x;
            ws.bin_entries[ws.bin_start[bin] + atomicAdd(&ws.bin_count[bin], 1)] = tri;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Primary rays, face by face.  The pixel-by-pixel pass above makes every pixel of a bin run through all faces that
// touch the bin -- a dozen exact tests per pixel where a face covers two or three pixels, and as long as the longest
// list for the launch as a whole.  Turned around, every face is tested against the pixels of its own projected box
// only (six pairs per face on the 1080p bunny, a sixth of the tests) and the pixels keep the best candidate through
// one 64-bit atomic maximum per accepted pair: the complement of (distance bits, inverted rank), so that zero means
// "nothing yet", the nearest face wins and equal distances go to the face the reference meets later
// (src/flyscene.cpp:760).  No bins, no lists, no second pass over the faces.
// A wave takes 64 faces.  Where all their boxes are tiny (a distant or finely tessellated mesh) every lane steps through
// its own face's pixels; otherwise the wave's (face, pixel) pairs are numbered consecutively and dealt to the lanes 64
// at a time, whatever the sizes of the boxes, with the faces' records staged in LDS.  Faces whose box exceeds
// kHugePixels (or is unbounded: a corner beside or behind the eye) go to the list that the compaction kernel tests
// against every pixel; when that list grows long (the eye inside a dense mesh) the pass gives way to the tree walk.
// ---------------------------------------------------------------------------------------------
#ifndef RT_CLUSTER_CULL
#define RT_CLUSTER_CULL(fc) ((fc).num_parts > 1)
#endif
constexpr int kHugePixels = 16384; // a face with a larger pixel box is tested pixel by pixel in the compaction kernel

// Can this part skip the 64 faces of record group `cluster` without looking at them?  Yes when the group's box lies wholly
// in front of the eye and the rows its projection covers all belong to other parts (or none is on the screen): the
// projection of a convex body in front of the eye plane lies inside the hull of its projected corners, so every face of
// the group would find the same in its own test in face_raster_kernel.  Every lane of the wave must call this; the
// eight corners are projected by lanes 0..7 of every group of eight, and all lanes return the same answer.
// Rounding: with the depth of every corner at least a fiftieth of its reach, a projected corner is within a quarter of a
// pixel of its true position at 8K (the bound at RT_BIN_MARGIN with R / s <= 50) and within 0.9 pixel at the largest
// image the face pass takes; the rows are widened by the margin and two pixels.
__device__ __forceinline__ bool cluster_outside_part(const DeviceScene &sc, const FrameConst &fc, int cluster, int lane) {
    const float4 lo = sc.cluster_box[2 * cluster], hi = sc.cluster_box[2 * cluster + 1];
    const V3 p = mk((lane & 1) ? hi.x : lo.x, (lane & 2) ? hi.y : lo.y, (lane & 4) ? hi.z : lo.z);
    const V3 r = p - mk(fc.origin[0], fc.origin[1], fc.origin[2]);
    const float *m = fc.lin_inv;
    const float qx = m[0] * r.x + (m[1] * r.y + m[2] * r.z);
    const float qy = m[3] * r.x + (m[4] * r.y + m[5] * r.z);
    const float qz = m[6] * r.x + (m[7] * r.y + m[8] * r.z);
    const float s = qz * fc.proj_s; // as in project_point
    const float reach = fabsf(qx) + fabsf(qy) + fabsf(qz);
    float smin = s, rmax = reach;
    for (int off = 1; off < 8; off <<= 1) smin = fminf(smin, __shfl_xor(smin, off, 64)), rmax = fmaxf(rmax, __shfl_xor(rmax, off, 64));
    if (!(smin > 0.02f * rmax)) return false; // a corner near or behind the eye plane (or not a number): no verdict
    const float rs = __builtin_amdgcn_rcpf(s);
    const float nx = qx * rs - fc.cam_e[0], ny = qy * rs - fc.cam_e[1];
    const float px = fc.px_off + nx * fc.px_scale;
    const float py = fc.py_off - ny * fc.py_scale;
    float xlo = px, xhi = px, ylo = py, yhi = py;
    for (int off = 1; off < 8; off <<= 1) {
        xlo = fminf(xlo, __shfl_xor(xlo, off, 64)), xhi = fmaxf(xhi, __shfl_xor(xhi, off, 64));
        ylo = fminf(ylo, __shfl_xor(ylo, off, 64)), yhi = fmaxf(yhi, __shfl_xor(yhi, off, 64));
    }
    const float lim = 1e6f;
    if (!(fabsf(xlo) < lim && fabsf(xhi) < lim && fabsf(ylo) < lim && fabsf(yhi) < lim)) return false; // (also: not a number)
    float spread = fabsf(rs) * (1.0f + fabsf(qx * rs) + fabsf(qy * rs)); // as in project_point
    for (int off = 1; off < 8; off <<= 1) spread = fmaxf(spread, __shfl_xor(spread, off, 64));
    const float slack = fc.raster_margin + fc.coord_slack * spread + 2.0f;
    const float fx0 = fmaxf(ceilf(xlo - slack - fc.sub_x), 0.0f), fy0 = fmaxf(ceilf(ylo - slack - fc.sub_y), 0.0f);
    const float fx1 = fminf(floorf(xhi + slack - fc.sub_x), (float)(fc.width - 1)), fy1 = fminf(floorf(yhi + slack - fc.sub_y), (float)(fc.height - 1));
    if (fx0 > fx1 || fy0 > fy1) return true; // off the screen
    const int t0 = (int)fast_div((unsigned)(int)fy0, (unsigned)fc.row_tile, fc.row_tile_magic);
    const int t1 = (int)fast_div((unsigned)(int)fy1, (unsigned)fc.row_tile, fc.row_tile_magic);
    if (t1 - t0 + 1 >= fc.num_parts) return false;
    for (int t = t0; t <= t1; ++t) {
        const int lt = (int)fast_div((unsigned)t, (unsigned)fc.num_parts, fc.num_parts_magic);
        if (t - lt * fc.num_parts == fc.part) return false;
    }
    return true;
}

// Number of this part's rows among the image rows [0, y): the local row of image row y if the part renders it.
__device__ __forceinline__ int local_rows_below(const FrameConst &fc, int y) {
    if (fc.num_parts == 1) return y;
    const int tile = (int)fast_div((unsigned)y, (unsigned)fc.row_tile, fc.row_tile_magic);
    const int turn = (int)fast_div((unsigned)tile, (unsigned)fc.num_parts, fc.num_parts_magic);
    const int who = tile - turn * fc.num_parts;
    return turn * fc.row_tile + (who > fc.part ? fc.row_tile : who == fc.part ? y - tile * fc.row_tile : 0);
}

// (Face, pixel) pairs: the exact plane + three-edge test of src/flyscene.cpp:752-789 for the primary ray of the pixel in column x
// of LOCAL row lr.  Several pairs of one lane side by side: the table entries of all their pixels in one round trip, then all
// keys in the next (taken one after the other, a lane's pairs are a chain of two dependent fetches each).
struct FaceRec {
    float4 pn, r0, r1, r2;
};
template <bool COUNT, int U>
__device__ __forceinline__ void raster_pairs(const FrameConst &fc, const Workspace &ws, const FaceRec (&f)[U], const int (&x)[U], const int (&lr)[U],
                                             const bool (&on)[U], unsigned &w_plane, unsigned &w_edge) {
    const V3 o = mk(fc.origin[0], fc.origin[1], fc.origin[2]);
    float nx[U], ny[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        nx[u] = ny[u] = 0.0f;
        if (on[u]) nx[u] = ws.dir_x[x[u]], ny[u] = ws.dir_y[lr[u]];
    }
#if RT_KEEP_LOADS
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(nx[u]), "+v"(ny[u]));
#endif
    V3 d[U];
    float t[U];
    bool go[U];
    unsigned long long seen[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        d[u] = primary_direction_of(fc, nx[u], ny[u]);
        if (COUNT) w_plane += (unsigned)on[u];
        const V3 n = mk(f[u].pn.x, f[u].pn.y, f[u].pn.z);
        t[u] = (f[u].pn.w - dot(o, n)) / dot(d[u], n);
        go[u] = on[u] && (t[u] >= fc.t_eps && t[u] <= FLT_MAX); // also rejects NaN (see file header)
    }
    // what the pixels hold already (possibly a moment ago: a key only ever gets nearer) spares the edge tests of faces behind it
#pragma unroll
    for (int u = 0; u < U; ++u) {
        seen[u] = 0ull;
        if (go[u]) seen[u] = ws.pix_key[(size_t)lr[u] * (size_t)fc.width + (size_t)x[u]];
    }
#if RT_KEEP_LOADS
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(seen[u]));
#endif
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!go[u] || (seen[u] != 0ull && t[u] > __int_as_float((int)(unsigned)(~seen[u] >> 32)))) continue;
        if (COUNT) ++w_edge;
        const V3 n = mk(f[u].pn.x, f[u].pn.y, f[u].pn.z);
        const V3 v0 = mk(f[u].r0.x, f[u].r0.y, f[u].r0.z), v1 = mk(f[u].r1.x, f[u].r1.y, f[u].r1.z), v2 = mk(f[u].r2.x, f[u].r2.y, f[u].r2.z);
        const V3 hp = o + t[u] * d[u];
        if (dot(cross(v1 - v0, hp - v0), n) < 0) continue;
        if (dot(cross(v2 - v1, hp - v1), n) < 0) continue;
        if (dot(cross(v0 - v2, hp - v2), n) < 0) continue;
        atomicMax(ws.pix_key + ((size_t)lr[u] * (size_t)fc.width + (size_t)x[u]), ~hit_key(t[u], __float_as_int(f[u].r1.w)));
        if (ws.prefilled) ws.tile_flag[(lr[u] >> 3) * fc.tiles_x + (x[u] >> 3)] = 1; // the compaction kernel then looks at flagged tiles only
    }
}

// What a primary ray that hits nothing leaves in the output planes (background colour, RTAMD_FACE_MISS, distance 0:
// src/flyscene.cpp:650-658), written into EVERY pixel in whole coalesced lines by threads i, i + stride, ...
__device__ __forceinline__ void prefill_planes(const FrameConst &fc, float *rgb, int32_t *face_id, float *t_hit, int i, int stride) {
    const int pixels = fc.local_rows * fc.width;
    // four floats a store where the plane allows it: float 4 k + e holds channel (k + e) mod 3
    // (selected component by component: a choice between whole float4 values is lowered to an array in scratch memory)
    const int quads = ((uintptr_t)rgb & 15) == 0 ? (3 * pixels) >> 2 : 0;
    const float b0 = fc.bg[0], b1 = fc.bg[1], b2 = fc.bg[2];
    for (int k = i; k < quads; k += stride) {
        const int m = k - (k / 3) * 3;
        float4 v;
        v.x = m == 0 ? b0 : m == 1 ? b1 : b2;
        v.y = m == 0 ? b1 : m == 1 ? b2 : b0;
        v.z = m == 0 ? b2 : m == 1 ? b0 : b1;
        v.w = v.x;
        ((float4 *)rgb)[k] = v;
    }
    for (int k = 4 * quads + i; k < 3 * pixels; k += stride) {
        const int c = k - (k / 3) * 3;
        rgb[k] = c == 0 ? fc.bg[0] : c == 1 ? fc.bg[1] : fc.bg[2];
    }
    if (face_id)
        for (int k = i; k < pixels; k += stride) face_id[k] = -2;
    if (t_hit)
        for (int k = i; k < pixels; k += stride) t_hit[k] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// A screen-space test in front of the exact one.  The box of a projected triangle holds about twice
// the pixels the triangle covers, and the exact test of a (face, pixel) pair costs some 270 instructions (the pixel's
// direction with its square root and division, the plane's division, three cross products).  Three edge functions of
// the PROJECTED triangle, widened by the margin the boxes are widened by, tell for a dozen instructions that a pixel's
// sample lies outside for good; the pairs that survive are collected in a queue of the wave and go through the exact test
// 128 at a time, all lanes busy.  Only pairs the exact test would reject are dropped: same keys, same image.
// ---------------------------------------------------------------------------------------------
// G_k(x', y') = a_k x' + b_k y' + c_k for pixel (first column + x', first row + y') of the face's box: negative for some
// k = the sample lies outside the projected triangle by more than the margin.  E(r) = dx (r.y - ey) - dy (r.x - ex) is
// >= 0 left of the edge from e along (dx, dy); a point within the margin (in x and in y) of the triangle has
// E >= -margin (|dx| + |dy|) for all three edges.  The margin is the one of the boxes (four times the worst error of a
// projected corner plus the rounding of the exact test, see RT_BIN_MARGIN); a quarter more covers the rounding of G itself
// (coordinates below 2^14 and box-relative pixels: at most a sixth of the margin term).  A face whose orientation on the
// screen is not certain from its rounded area, or with a corner far off the screen, keeps all pixels of its box.
struct EdgeFilter {
    float a[3], b[3], c[3];
};
__device__ __forceinline__ EdgeFilter edge_filter(const FrameConst &fc, const ProjectedFace &f) {
    EdgeFilter e;
#pragma unroll
    for (int k = 0; k < 3; ++k) e.a[k] = 0.0f, e.b[k] = 0.0f, e.c[k] = 1.0f;
    if (f.empty || f.unbounded) return e;
    const float lim = 16384.0f;
    const bool near_screen = fabsf(f.x0) < lim && fabsf(f.y0) < lim && fabsf(f.x1) < lim && fabsf(f.y1) < lim && fabsf(f.x2) < lim && fabsf(f.y2) < lim;
    const float ux = f.x1 - f.x0, uy = f.y1 - f.y0, vx = f.x2 - f.x0, vy = f.y2 - f.y0;
    const float area = ux * vy - uy * vx;
    const float area_slop = 1e-3f + 4e-6f * (fabsf(ux * vy) + fabsf(uy * vx)); // (sixteen times the rounding of the expression above)
    if (!near_screen || !(fabsf(area) > area_slop)) return e;
    // corners relative to the sample of the box's first pixel
    const float ox = (float)f.px0 + fc.sub_x, oy = (float)f.py0 + fc.sub_y;
    const float px[3] = {f.x0 - ox, f.x1 - ox, f.x2 - ox}, py[3] = {f.y0 - oy, f.y1 - oy, f.y2 - oy};
    const float margin = 1.25f * f.margin;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int n = k == 2 ? 0 : k + 1;
        // counter-clockwise (area > 0): inside is left of p_k -> p_n; clockwise: left of p_n -> p_k
        const int from = area > 0.0f ? k : n, to = area > 0.0f ? n : k;
        const float dx = px[to] - px[from], dy = py[to] - py[from];
        e.a[k] = -dy, e.b[k] = dx;
        e.c[k] = (dy * px[from] - dx * py[from]) + margin * (fabsf(dx) + fabsf(dy));
    }
    return e;
}

#ifndef RT_FILTER_PAIRS
#define RT_FILTER_PAIRS 2
#endif
// (a step adds at most 64 RT_FILTER_PAIRS entries to fewer than 64 RT_RASTER_PAIRS waiting ones)
constexpr int kFilterRing = 64 * (RT_FILTER_PAIRS + RT_RASTER_PAIRS) <= 256 ? 256 : 512;
static_assert(64 * (RT_FILTER_PAIRS + RT_RASTER_PAIRS) <= kFilterRing, "ring too small");
struct RasterWaveLds {
    float4 rec[4][64]; // the four quarters of the wave's face records
    int4 box[64];      // first column | first LOCAL row << 16, box width | first IMAGE row << 16, floor(2^32 / width), number of the face's first pair
    float edge[9][64]; // EdgeFilter
    int first[64];     // number of the face's first pair again, densely (the search reads nothing else)
    int queue[kFilterRing]; // ring of pairs that passed the screen-space test: face (6 bits) | number of the pixel in its box << 6
};

// the exact test for `n` <= 64 RT_RASTER_PAIRS pairs of the queue, starting at ring position `head`
template <bool COUNT>
__device__ __forceinline__ void raster_queued(const FrameConst &fc, const Workspace &ws, const RasterWaveLds &s, int head, int n, int lane, unsigned &w_plane,
                                              unsigned &w_edge) {
    FaceRec rec[RT_RASTER_PAIRS];
    int px[RT_RASTER_PAIRS], pr[RT_RASTER_PAIRS];
    bool on[RT_RASTER_PAIRS];
#pragma unroll
    for (int u = 0; u < RT_RASTER_PAIRS; ++u) {
        const int at = 64 * u + lane;
        on[u] = at < n;
        const int e = s.queue[(head + (on[u] ? at : 0)) & (kFilterRing - 1)];
        const int lo = e & 63;
        const unsigned k = (unsigned)e >> 6;
        const int4 box = s.box[lo];
        const unsigned w = (unsigned)box.y & 0xffffu;
        const unsigned row = fast_div(k, w, (unsigned)box.z);
        px[u] = (box.x & 0xffff) + (int)(k - row * w), pr[u] = (int)((unsigned)box.x >> 16) + (int)row;
        rec[u].pn = s.rec[0][lo], rec[u].r0 = s.rec[1][lo], rec[u].r1 = s.rec[2][lo], rec[u].r2 = s.rec[3][lo];
    }
    raster_pairs<COUNT, RT_RASTER_PAIRS>(fc, ws, rec, px, pr, on, w_plane, w_edge);
}

#ifndef RT_FILTER_WAVES
#define RT_FILTER_WAVES 1
#endif
// Waves per workgroup: one (1.1 M faces at 4K: 88.9 us; two 91.0, four 95.1 -- a workgroup leaves when its slowest wave does).
constexpr int kFilterWaves = RT_FILTER_WAVES;
// `fshift`: a wave takes 1 << fshift faces (6: one per lane).  A mesh of few faces makes few waves -- 69 K faces are 1085
// waves of 64, a seventh of what the chip holds, each with hundreds of pairs to step through one batch after the other -- so
// the launcher gives a wave fewer faces then (launch_primary_faces); their pairs are dealt over all 64 lanes either way.
// Workgroups from `raster_blocks` on do not look at faces: they write the miss values into the output planes (ws.prefill_with_faces:
// the sweep that prep_kernel otherwise makes, which nothing in this kernel depends on -- a launch of its own costs a frame that
// is alone on the GPU its duration and a kernel boundary).
template <bool COUNT>
__global__ __launch_bounds__(64 * kFilterWaves) void face_raster_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int num_tris, int fshift, int raster_blocks, float *rgb,
                                                         int32_t *face_id, float *t_hit) {
    if ((int)blockIdx.x >= raster_blocks) {
        prefill_planes(fc, rgb, face_id, t_hit, ((int)blockIdx.x - raster_blocks) * (64 * kFilterWaves) + (int)threadIdx.x,
                       ((int)gridDim.x - raster_blocks) * (64 * kFilterWaves));
        return;
    }
    __shared__ RasterWaveLds s_wave[kFilterWaves];
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    RasterWaveLds &s = s_wave[wave];
    const int wave_first = ((int)blockIdx.x * kFilterWaves + wave) << fshift;
    const int tri = wave_first + lane;
    const bool holds = lane < (1 << fshift) && tri < num_tris; // this lane holds a face
    unsigned w_box = 0, w_plane = 0, w_edge = 0;
    ProjectedFace f;
    f.empty = true, f.unbounded = false, f.px0 = f.py0 = 0, f.px1 = f.py1 = -1;
    f.x0 = f.y0 = f.x1 = f.y1 = f.x2 = f.y2 = 0.0f, f.margin = fc.raster_margin;
    // a rank that renders some rows only: groups of faces that lie in other ranks' rows are dropped unseen
    // (a wave's faces lie in one group of 64 records: 1 << fshift divides 64)
    bool skip = false;
    if (RT_CLUSTER_CULL(fc) && wave_first < num_tris) skip = cluster_outside_part(sc, fc, wave_first >> 6, lane);
    if (holds && !skip) f = project_face(sc, fc, tri);
    if (COUNT && holds && !skip) ++w_box; // faces projected
    int w = f.px1 - f.px0 + 1, h = f.py1 - f.py0 + 1;
    const bool large = f.unbounded || w * h > kHugePixels;
    // From here on the box's rows are this part's own (packed) rows: a face that spans the row tiles of several parts is
    // tested against the rows rendered here only, and one that touches none of them is dropped.
    int row0 = f.py0, first_row = f.py0; // (local and image row of the box's first row this part renders)
    if (!f.empty && fc.num_parts > 1) {
        row0 = local_rows_below(fc, f.py0);
        h = local_rows_below(fc, f.py1 + 1) - row0;
        if (h <= 0) f.empty = true;
    }
    const bool huge = !f.empty && large;
    if (huge) {
        const int at = atomicAdd(&ws.bin_state->num_big, 1);
        if (at < kMaxHugeFaces) ws.big[at] = make_int4(tri, f.px0 | (f.py0 << 16), f.px1 | (f.py1 << 16), 0);
    }
    const int count = (f.empty || huge) ? 0 : w * h;
    int incl = count; // inclusive scan of the counts over the wave
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int up = __shfl_up(incl, dlt, 64);
        if (lane >= dlt) incl += up;
    }
    const int total = __shfl(incl, 63, 64);
    if (total > 0) {
        const int mine = holds ? tri : wave_first; // (lanes without a face have no pairs; any record will do)
        s.rec[0][lane] = sc.tris[(size_t)mine * 4 + 0], s.rec[1][lane] = sc.tris[(size_t)mine * 4 + 1];
        s.rec[2][lane] = sc.tris[(size_t)mine * 4 + 2], s.rec[3][lane] = sc.tris[(size_t)mine * 4 + 3];
        s.box[lane] = make_int4(f.px0 | (row0 << 16), max(w, 1) | (first_row << 16), (int)(w <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned long long)w)),
                                incl - count);
        s.first[lane] = incl - count;
        const EdgeFilter ef = edge_filter(fc, f);
#pragma unroll
        for (int k = 0; k < 3; ++k) s.edge[k][lane] = ef.a[k], s.edge[3 + k][lane] = ef.b[k], s.edge[6 + k][lane] = ef.c[k];
        wave_sync();
        int head = 0, tail = 0; // ring positions (the wave's own: the same in all lanes)
        // RT_FILTER_PAIRS pairs per lane and step: their searches are independent chains of LDS reads that overlap
        for (int j0 = 0; j0 < total; j0 += 64 * RT_FILTER_PAIRS) {
            int lo[RT_FILTER_PAIRS], hi[RT_FILTER_PAIRS], jc[RT_FILTER_PAIRS];
            bool pass[RT_FILTER_PAIRS];
#pragma unroll
            for (int u = 0; u < RT_FILTER_PAIRS; ++u) {
                const int j = j0 + 64 * u + lane;
                pass[u] = j < total;
                jc[u] = pass[u] ? j : 0;
                lo[u] = 0, hi[u] = 63;
            }
            // the face of pair j: the last one whose first pair is <= j (faces without pixels share their successor's number)
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                int first[RT_FILTER_PAIRS], mid[RT_FILTER_PAIRS];
#pragma unroll
                for (int u = 0; u < RT_FILTER_PAIRS; ++u) mid[u] = (lo[u] + hi[u] + 1) >> 1, first[u] = s.first[mid[u]];
#pragma unroll
                for (int u = 0; u < RT_FILTER_PAIRS; ++u) {
                    if (first[u] <= jc[u])
                        lo[u] = mid[u];
                    else
                        hi[u] = mid[u] - 1;
                }
            }
            int entry[RT_FILTER_PAIRS];
#pragma unroll
            for (int u = 0; u < RT_FILTER_PAIRS; ++u) {
                const int4 box = s.box[lo[u]];
                const unsigned k = (unsigned)(jc[u] - box.w), bw = (unsigned)box.y & 0xffffu;
                const unsigned row = fast_div(k, bw, (unsigned)box.z);
                const float xr = (float)(k - row * bw);
                // rows of the box in the IMAGE: under a partition the part's rows are not consecutive there
                const int local_row = (int)((unsigned)box.x >> 16) + (int)row;
                const int image_row = fc.num_parts == 1 ? local_row : global_row(fc, local_row);
                const float yr = (float)(image_row - (int)((unsigned)box.y >> 16));
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const float g = __builtin_fmaf(s.edge[e][lo[u]], xr, __builtin_fmaf(s.edge[3 + e][lo[u]], yr, s.edge[6 + e][lo[u]]));
                    pass[u] = pass[u] && !(g < 0.0f);
                }
                entry[u] = lo[u] | (int)(k << 6);
            }
#pragma unroll
            for (int u = 0; u < RT_FILTER_PAIRS; ++u) {
                const unsigned long long passed = __ballot(pass[u]);
                if (pass[u]) s.queue[(tail + (int)__popcll(passed & ((1ull << lane) - 1ull))) & (kFilterRing - 1)] = entry[u];
                tail += (int)__popcll(passed);
            }
            wave_sync();
            while (tail - head >= 64 * RT_RASTER_PAIRS) {
                raster_queued<COUNT>(fc, ws, s, head, 64 * RT_RASTER_PAIRS, lane, w_plane, w_edge);
                head += 64 * RT_RASTER_PAIRS;
            }
        }
        if (tail > head) raster_queued<COUNT>(fc, ws, s, head, tail - head, lane, w_plane, w_edge);
    }
    if (COUNT) flush_work(w_box, w_plane, w_edge, ws.work + 2);
}

// Candidates per round trip in raster_kernel.  Two: 70 VGPRs and seven waves per SIMD; four take 84 and five waves,
// which costs more than the longer chains of dependent loads (bunny at 4K 6 %, 1.1 M-face field 3 %).  (Measured while the
// SLP vectoriser was on for this file; this pass runs only behind RTAMD_PRIMARY_RAYS_BINS and was not measured again.)
#ifndef RT_RASTER_BATCH
#define RT_RASTER_BATCH 2
#endif
template <bool COUNT>
__global__ __launch_bounds__(256) void raster_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int total_static) {
    if (!bins_accepted(ws)) return;
    const int slot = (int)(blockIdx.x * 256 + threadIdx.x);
    unsigned w_box = 0, w_plane = 0, w_edge = 0;
    if (slot < total_static) {
        int x, row;
        if (!slot_pixel(fc, slot, x, row)) {
            ws.slot_tri[slot] = -2; // padding of an edge tile
        } else {
            const int gy = global_row(fc, row);
            const int bx = x / kBinPixels, by = gy / kBinPixels;
            const int bin = by * fc.bins_x + bx;
            const int nlist = ws.bin_count[bin], nbig = ws.bin_state->num_big;
            if (COUNT) ++w_box;
            float tbest = FLT_MAX;
            int best_tri = -2, best_rank = -1;
            // most pixels of most frames see no face at all: no candidates, no ray
            if (nlist + nbig > 0 || sc.num_spheres > 0) {
            const int begin = ws.bin_start[bin];
            const V3 o = mk(fc.origin[0], fc.origin[1], fc.origin[2]);
            const V3 d = primary_direction(fc, ws, x, row);
            // intersects(): spheres first, then triangles, sharing the running maximum (:673-676)
            if (sc.num_spheres > 0 && spheres_hit(sc, o, d, fc.t_eps, tbest)) best_tri = -1;
            // Candidates are taken a few at a time (RT_RASTER_BATCH): their index loads, then their plane loads, are
            // independent, so a long list costs that much fewer dependent-load round trips.  The order of the candidates does
            // not matter: equal distances are settled by the faces' ranks.
            for (int e = 0; e < nlist + nbig; e += RT_RASTER_BATCH) {
                int idx[RT_RASTER_BATCH];
#pragma unroll
                for (int k = 0; k < RT_RASTER_BATCH; ++k) {
                    const int ek = e + k;
                    int i = -1;
                    if (ek < nlist) {
                        i = ws.bin_entries[begin + ek];
                    } else if (ek < nlist + nbig) {
                        const int4 g = ws.big[ek - nlist];
                        if (!(bx < (g.y & 0xffff) || bx > (g.z & 0xffff) || by < (g.y >> 16) || by > (g.z >> 16))) i = g.x;
                    }
                    idx[k] = i;
                }
                // The planes of the batch up front, the vertices only of a face whose plane the ray meets in range.  (Whole
                // records up front save that second round trip and make a lone frame's raster pass 5 % shorter, but
                // with frames in flight the fetches they waste cost 1-7 % of the throughput.)
                float4 pn[RT_RASTER_BATCH];
#pragma unroll
                for (int k = 0; k < RT_RASTER_BATCH; ++k) pn[k] = sc.tris[(size_t)max(idx[k], 0) * 4];
#pragma unroll
                for (int k = 0; k < RT_RASTER_BATCH; ++k) {
                    const int i = idx[k];
                    if (i < 0) continue;
                    // exact plane + three-edge test, src/flyscene.cpp:752-789
                    if (COUNT) ++w_plane;
                    const V3 n = mk(pn[k].x, pn[k].y, pn[k].z);
                    const float t = (pn[k].w - dot(o, n)) / dot(d, n);
                    if (!(t >= fc.t_eps && t <= tbest)) continue; // also rejects NaN (see file header)
                    const float4 *rec = sc.tris + (size_t)i * 4;
                    const float4 r0 = rec[1], r1 = rec[2], r2 = rec[3];
                    const int rank = __float_as_int(r1.w);
                    if (t == tbest && rank < best_rank) continue; // equal distance: the later face wins (:760)
                    if (COUNT) ++w_edge;
                    const V3 v0 = mk(r0.x, r0.y, r0.z), v1 = mk(r1.x, r1.y, r1.z), v2 = mk(r2.x, r2.y, r2.z);
                    const V3 hp = o + t * d;
                    if (dot(cross(v1 - v0, hp - v0), n) < 0) continue;
                    if (dot(cross(v2 - v1, hp - v1), n) < 0) continue;
                    if (dot(cross(v0 - v2, hp - v2), n) < 0) continue;
                    tbest = t;
                    best_tri = i;
                    best_rank = rank;
                }
            }
            }
            ws.slot_tri[slot] = best_tri;
            ws.slot_t[slot] = tbest;
        }
    }
    if (COUNT) flush_work(w_box, w_plane, w_edge, ws.work + 2);
}

// ---------------------------------------------------------------------------------------------
// Shadow rays through the candidate grids (shadow_grid.hpp).  All shadow rays of one light sample are
// parallel (their direction is the sample's position, src/flyscene.cpp:884), so the faces that can occlude a
// point are found by projecting the point along that direction into a 2D grid built once per scene.  One wave
// = 64 consecutive hits x one sample; a lane looks up its cell and runs the exact plane + three-edge test
// (src/flyscene.cpp:752-789, same arithmetic as the tree walk) over the cell's list, which is sorted by the
// faces' farthest extent along the direction: the scan ends at the first face that lies wholly behind the
// ray's origin, or at the first face that blocks the sample.
// ---------------------------------------------------------------------------------------------
// Candidates per round trip in shadow_grid_kernel.  Four: 62 VGPRs, still eight waves per SIMD (three: 54 registers and one
// more trip through the loop for many waves, 1.5 % slower; with the packed arithmetic of the SLP vectoriser, which the
// Makefile turns off, four cost 80 registers and lost to three).
#ifndef RT_SHADOW_BATCH
#define RT_SHADOW_BATCH 4
#endif
static_assert(RT_SHADOW_BATCH >= 1 && RT_SHADOW_BATCH <= 8, "the entry arrays are padded by seven entries (kShadowGridPadding in shadow_grid.hpp)");
#ifndef RT_SHADOW_SCALAR_HEADERS
#define RT_SHADOW_SCALAR_HEADERS 1
#endif
#ifndef RT_SHADOW_XCD
#define RT_SHADOW_XCD 1
#endif
#ifndef RT_SHADOW_CORNERS_AHEAD
#define RT_SHADOW_CORNERS_AHEAD 0
#endif
template <bool COUNT>
__device__ __forceinline__ void shadow_grid_body(const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, int block, int nblocks) {
    const int lane = (int)threadIdx.x;
    const int nshadow = sc.num_lights * sc.spl;
    const int nhits = ws.stage[stage].hits;
    // Lanes per ray.  A launch too small to fill the chip (the shadow rays of bounce hits) is bound by the ray
    // with the longest candidate list; sixteen lanes then share one ray, each taking every sixteenth batch of
    // its list, and combine their verdicts at the end.
    const long long rays = (long long)nhits * nshadow;
    const int split_shift = rays < RT_SHADOW_SPLIT_BELOW ? RT_SHADOW_SPLIT_SHIFT : 0;
    const int split = 1 << split_shift, per_wave_shift = 6 - split_shift;
    const int sub = lane & (split - 1);
    const int chunks = (nhits + (1 << per_wave_shift) - 1) >> per_wave_shift;
    const int ngroups = chunks * nshadow;
    unsigned w_box = 0, w_plane = 0, w_edge = 0;
    unsigned shot = 0; // shadow rays this wave has shot
    // Workgroups b and b + 8 run on the same XCD and share its 4 MiB L2.  Dealt out in launch order, every XCD would see
    // every eighth group of every sample, fetch the lines that neighbouring groups share once per XCD, and stream all
    // grids past its L2.  Instead the XCD class of a workgroup (b & 7) takes runs of consecutive groups of the
    // sample-major sequence.  With a dozen samples or more the run is a whole eighth of the sequence: the grids of one or
    // two samples, which the L2 holds while the class works through them row of tiles after row of tiles (1080p bunny, 13
    // samples: shadow launches 6 % shorter, frames in flight 3 % faster).  With fewer samples an eighth is a band of the
    // image or a single light, whose costs differ too much for one launch to balance (1.1 M-face field at 4K, one
    // sample: 5 % slower); runs of 64 groups keep the lines shared by neighbours and the balance (1 % faster).
    const bool by_xcd = RT_SHADOW_XCD != 0 && (nblocks & 7) == 0;
    const int run = nshadow >= 12 ? max((ngroups + 7) >> 3, 1) : 64;
    const int per_xcd = ((ngroups + 8 * run - 1) / (8 * run)) * run;
    for (int g = block; by_xcd ? (g >> 3) < per_xcd : g < ngroups; g += nblocks) {
        const int j = g >> 3, turn = j / run;
        const int group = by_xcd ? turn * (8 * run) + (g & 7) * run + (j - turn * run) : g;
        if (group >= ngroups) continue;
        // sample-major: the waves in flight at one time work on one grid, not on all of them (8 lights at 8K: -12 %; the
        // thirteen samples of a soft shadow, whose candidates overlap, gain nothing from hit-major order either)
        const int sample = group / chunks, chunk = group - sample * chunks;
        const int hit = (chunk << per_wave_shift) + (lane >> split_shift);
        const bool valid = hit < nhits;
        shot += (unsigned)__popcll(__ballot(valid && sub == 0)); // (sixteen lanes may share a ray: it counts once)
        const float *hp = ws.hits[depth & 1].p;
        const int hit_c = valid ? hit : 0; // lanes past the end stay in step for the cross-lane vote below
        const V3 o = mk(hp[hit_c], hp[(size_t)ws.capacity + hit_c], hp[2 * (size_t)ws.capacity + hit_c]);
#if RT_SHADOW_SCALAR_HEADERS
        // (the sample is the same for the whole wave: its direction and its grid header come through the scalar unit, from
        // arrays no kernel ever writes, instead of as sixty-four identical vector fetches)
        typedef const float __attribute__((address_space(4))) *ConstFloats;
        const ConstFloats sample_dir = (ConstFloats)(sc.samples + 3 * __builtin_amdgcn_readfirstlane(sample));
        const V3 d = mk(sample_dir[0], sample_dir[1], sample_dir[2]);
#else
        const V3 d = mk(sc.samples[3 * sample], sc.samples[3 * sample + 1], sc.samples[3 * sample + 2]);
#endif
        bool blocked = false;
        // intersects(): spheres first (:673-676)
        float tbest = FLT_MAX;
        if (sc.num_spheres > 0 && spheres_hit(sc, o, d, fc.t_eps, tbest)) blocked = true;
        if (!blocked && valid) {
#if RT_SHADOW_SCALAR_HEADERS
            const ConstFloats hdr = (ConstFloats)((const float *)sc.grid_headers + 16 * __builtin_amdgcn_readfirstlane(sample));
            const float4 g0 = make_float4(hdr[0], hdr[1], hdr[2], hdr[3]), g1 = make_float4(hdr[4], hdr[5], hdr[6], hdr[7]);
            const float4 g2 = make_float4(hdr[8], hdr[9], hdr[10], hdr[11]), g3 = make_float4(hdr[12], hdr[13], hdr[14], hdr[15]);
#else
            const float4 g0 = sc.grid_headers[sample * 4 + 0], g1 = sc.grid_headers[sample * 4 + 1];
            const float4 g2 = sc.grid_headers[sample * 4 + 2], g3 = sc.grid_headers[sample * 4 + 3];
#endif
            // header: U (g0.xyz), V (g0.w g1.xy), W (g1.zw g2.x), u0 v0 inv_cell (g2.yzw), res first_cell (g3.xy)
            const float u = o.x * g0.x + (o.y * g0.y + o.z * g0.z);
            const float v = o.x * g0.w + (o.y * g1.x + o.z * g1.y);
            const float z = o.x * g1.z + (o.y * g1.w + o.z * g2.x);
            const float fx = floorf((u - g2.y) * g2.w), fy = floorf((v - g2.z) * g2.w);
            const int res = __float_as_int(g3.x), first_cell = __float_as_int(g3.y);
            if (COUNT && sub == 0) ++w_box;
            if (fx >= 0.0f && fy >= 0.0f && fx < (float)res && fy < (float)res) {
                const int cell = first_cell + (int)fy * res + (int)fx;
                const int begin = sc.grid_cells[cell], end = sc.grid_cells[cell + 1];
                // a few candidates at a time (RT_SHADOW_BATCH): independent entry loads, then independent plane loads
                for (int e = begin + RT_SHADOW_BATCH * sub; e < end && !blocked; e += RT_SHADOW_BATCH * split) {
                    // (one address for the batch: the entry array is padded, so the fetch may run past the end of the list,
                    // into the next list's entries, which are skipped below; records by 32-bit byte offsets from the base)
                    const int2 *const ep = sc.grid_entries + e;
                    int2 ent[RT_SHADOW_BATCH];
#pragma unroll
                    for (int k = 0; k < RT_SHADOW_BATCH; ++k) ent[k] = ep[k];
                    const char *const records = (const char *)sc.tris;
                    float4 pn[RT_SHADOW_BATCH];
                    // (a lane asks only for the planes it will test: entries past its list's end or behind its origin are masked out)
#pragma unroll
                    for (int k = 0; k < RT_SHADOW_BATCH; ++k) {
                        pn[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if (e + k < end && !(__int_as_float(ent[k].y) < z)) pn[k] = *(const float4 *)((const char *)sc.planes + ((unsigned)ent[k].x << 4));
                    }
#if RT_SHADOW_CORNERS_AHEAD
                    // Diagnostic build only (never the default): the corners of the batch's first candidates fetched together
                    // with the planes, whatever the lane's list length and whatever the plane test will say -- the variant that
                    // once ended in a memory fault.  Every record index such a fetch is formed from is checked against the
                    // number of records first, also for lanes past their list's end; a bad one is reported and skipped.
                    float4 ahead[RT_SHADOW_CORNERS_AHEAD][3];
#pragma unroll
                    for (int k = 0; k < RT_SHADOW_CORNERS_AHEAD; ++k) {
                        const unsigned idx = (unsigned)ent[k].x;
                        if (idx >= (unsigned)sc.num_records) {
                            printf("rtamd diagnostic: record index %u of %d formed from entry %d (+%d) of a list [%d, %d), lane %d sub %d\n", idx, sc.num_records, e, k, begin,
                                   end, lane, sub);
                            ahead[k][0] = ahead[k][1] = ahead[k][2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            continue;
                        }
                        const float4 *const rec = (const float4 *)(records + (idx << 6));
                        ahead[k][0] = rec[1], ahead[k][1] = rec[2], ahead[k][2] = rec[3];
                        keep(ahead[k][0]), keep(ahead[k][1]), keep(ahead[k][2]);
                    }
#endif
                    bool behind = false;
#pragma unroll
                    for (int k = 0; k < RT_SHADOW_BATCH; ++k) {
                        if (blocked || behind || e + k >= end) continue;
                        if (__int_as_float(ent[k].y) < z) { // this face and all later ones lie behind the origin
                            behind = true;
                            continue;
                        }
                        const int i = ent[k].x;
                        if (COUNT) ++w_plane;
                        const V3 n = mk(pn[k].x, pn[k].y, pn[k].z);
                        const float t = (pn[k].w - dot(o, n)) / dot(d, n);
                        if (!(t >= fc.t_eps && t <= tbest)) continue; // also rejects NaN (see file header)
                        const float4 *const rec = (const float4 *)(records + ((unsigned)i << 6));
#if RT_SHADOW_CORNERS_AHEAD
                        const float4 p0 = k < RT_SHADOW_CORNERS_AHEAD ? ahead[k < RT_SHADOW_CORNERS_AHEAD ? k : 0][0] : rec[1];
                        const float4 p1 = k < RT_SHADOW_CORNERS_AHEAD ? ahead[k < RT_SHADOW_CORNERS_AHEAD ? k : 0][1] : rec[2];
                        const float4 p2 = k < RT_SHADOW_CORNERS_AHEAD ? ahead[k < RT_SHADOW_CORNERS_AHEAD ? k : 0][2] : rec[3];
#else
                        const float4 p0 = rec[1];
                        const float4 p1 = rec[2];
                        const float4 p2 = rec[3];
#endif
                        if (COUNT) ++w_edge;
                        const V3 v0 = mk(p0.x, p0.y, p0.z), v1 = mk(p1.x, p1.y, p1.z), v2 = mk(p2.x, p2.y, p2.z);
                        const V3 at = o + t * d;
                        // the three edge functions side by side and ONE branch: tested one after the other, the compiler
                        // fetches the third corner only behind the first test, a second round trip per candidate
                        const float e0 = dot(cross(v1 - v0, at - v0), n);
                        const float e1 = dot(cross(v2 - v1, at - v1), n);
                        const float e2 = dot(cross(v0 - v2, at - v2), n);
                        if ((e0 < 0) | (e1 < 0) | (e2 < 0)) continue;
                        blocked = true;
                    }
                    if (behind) break;
                }
            }
        }
        // any lane of the ray's group found an occluder?  (every lane of the wave reaches this point)
        int any = blocked ? 1 : 0;
        for (int m = 1; m < split; m <<= 1) any |= __shfl_xor(any, m, 64);
        if (valid && sub == 0) ws.clear[depth & 1][sample * ws.capacity + hit] = any ? 0 : 1;
    }
    if (lane == 0 && shot != 0) atomicAdd(ws.shadow_shot + (block & (kShotSlots - 1)) * kShotStride, (unsigned long long)shot);
    if (COUNT) flush_work(w_box, w_plane, w_edge, ws.work + 1);
}


// (eight waves per SIMD need the kernel's scalar registers to stay within 96 as well: the wave-uniform grid header and scene
// pointers live there)
#ifndef RT_SHADOW_MIN_WAVES
#define RT_SHADOW_MIN_WAVES RT_WAVES_PER_SIMD
#endif
template <bool COUNT>
__global__ __launch_bounds__(64, RT_SHADOW_MIN_WAVES) void shadow_grid_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int stage, int depth) {
    shadow_grid_body<COUNT>(sc, fc, ws, stage, depth, (int)blockIdx.x, (int)gridDim.x);
}

// Two independent pieces of a frame in one launch: the shadow rays of recursion level `depth` and the closest hits of
// the reflection rays it spawned (level depth + 1).  Both need only the compaction of level `depth`; launched one
// after the other each would hold one of the few kernels the GPU runs at a time for as long as its slowest wave, with
// most of the chip idle behind it.  The first `walk_blocks` workgroups walk (their long chains start first), the rest
// cast shadow rays.
template <bool COUNT>
__global__ __launch_bounds__(64, RT_WAVES_PER_SIMD) void shadow_walk_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int stage, int depth,
                                                                            int walk_blocks) {
    __shared__ WalkLds lds;
    if ((int)blockIdx.x < walk_blocks)
        walk_steal_body<kQueue, COUNT>(sc, fc, ws, stage + 1, depth + 1, 0, (int)blockIdx.x, walk_blocks, lds);
    else
        shadow_grid_body<COUNT>(sc, fc, ws, stage, depth, (int)blockIdx.x - walk_blocks, (int)gridDim.x - walk_blocks);
}

// ---------------------------------------------------------------------------------------------
// Compaction: ray slots that hit something -> dense hit list (slot order inside a workgroup, one atomic
// per workgroup and list: atomics on one address serialise in L2, so a workgroup covers up to 2048 slots).  Also writes the hit point, the primary planes (face id, distance), the colour
// of primary rays that hit nothing (src/flyscene.cpp:650-658), and -- the reflection set-up of traceRay,
// src/flyscene.cpp:617-646 -- geometry and pixel of the reflection ray into the next queue.  The ray's
// throughput needs the light intensity and is filled in later by the shading kernel, through `child`.
// ---------------------------------------------------------------------------------------------
// kCompactSegments = 256-slot segments per workgroup (one atomic per list and workgroup): many for large
// images, where the atomics would dominate, few for small ones, which need the workgroups to fill the chip.
template <bool PRIMARY, int kCompactSegments, bool KEEP = false>
__device__ __forceinline__ void compact_body(const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, int spawn, int list_misses,
                                             float *rgb, int32_t *face_id, float *t_hit, int total_static, int block, int nblocks) {
    __shared__ int s_count[3][kCompactSegments][4]; // list (hits, spawned rays, misses) x segment x wave
    __shared__ int s_base[3];
    const int total = PRIMARY ? total_static : ws.stage[stage - 1].next;
    const HitList hl = ws.hits[depth & 1];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool from_keys = PRIMARY && ws.pix_key != nullptr && keys_accepted(ws); // the pass was resolved face by face
    // (faces too large for the face pass: counted by that pass, constant here -- read once, through the scalar unit, instead of
    // as a vector fetch with a wait of its own before every use)
    int nbig_listed = 0;
    if (PRIMARY && ws.pix_key != nullptr)
        nbig_listed = __builtin_amdgcn_readfirstlane(*(const int __attribute__((address_space(4))) *)&ws.bin_state->num_big);
    if (PRIMARY && ws.pix_key != nullptr && block == 0 && tid == 0 && ws.bin_feedback) { // the pass's verdict, for the next frames of the view
        ws.bin_feedback[0] = 0, ws.bin_feedback[1] = ws.bin_state->num_big; // [1]: faces too large for the face pass
        ws.bin_feedback[2] = from_keys ? 0 : 1, ws.bin_feedback[3] = 1;
    }
    // A few listed faces (a ground plane under a mesh is two): their records and boxes once per workgroup into LDS, instead of
    // three dependent fetches -- list entry, plane, corners -- per pixel and face in the loop below.
    constexpr int kBigLds = 8;
    __shared__ float4 s_big_rec[kBigLds][4];
    __shared__ int4 s_big_box[kBigLds];
    const bool big_in_lds = PRIMARY && from_keys && nbig_listed > 0 && nbig_listed <= kBigLds;
    if (big_in_lds) {
        if (tid < nbig_listed * 4) s_big_rec[tid >> 2][tid & 3] = sc.tris[(size_t)ws.big[tid >> 2].x * 4 + (size_t)(tid & 3)];
        if (tid < nbig_listed) s_big_box[tid] = ws.big[tid];
        __syncthreads();
    }
    // the grid may be smaller than the work (it is sized from what earlier frames of the view needed): stride over it
    for (int first = block * (256 * kCompactSegments); first < total; first += nblocks * (256 * kCompactSegments)) {
    // A run of tiles none of which the face pass flagged holds no hit, and its pixels already hold the miss values: the
    // workgroup moves on (the same decision in every thread; no LDS was touched, so the barrier below is not needed).
    if (PRIMARY && from_keys && ws.prefilled && nbig_listed == 0 && sc.num_spheres == 0) {
        int any = 0;
#pragma unroll
        for (int j = 0; j < kCompactSegments; ++j) {
            const int slot = first + j * 256 + tid;
            if (slot < total) any |= ws.tile_flag[slot >> 6];
        }
        if (!__syncthreads_or(any)) continue;
    }

    // ---- pass 1: classify the slots of every segment, count per wave
    int tri_of[kCompactSegments];
    float t_of[kCompactSegments];
    // KEEP: the pixel's direction, where pass 1 needed it (listed faces or spheres to test), is kept for pass 2 instead of being
    // worked out again (a dozen registers a thread: the launcher asks for it where earlier frames of the view had such faces)
    V3 d_of[kCompactSegments];
    const bool have_d = KEEP && PRIMARY && from_keys && (nbig_listed > 0 || sc.num_spheres > 0);
    unsigned hit_bits = 0, go_bits = 0, miss_bits = 0;
#pragma unroll
    for (int j = 0; j < kCompactSegments; ++j) {
        const int slot = first + j * 256 + tid;
        bool valid = slot < total;
        int x = 0, row = 0;
        if (PRIMARY) valid = slot_pixel_of_wave(fc, slot, x, row) && valid; // (`first` and the segments are multiples of 256)
        int tri = -2;
        float t_found = 0.0f;
        if (KEEP) d_of[j] = mk(0.0f, 0.0f, 1.0f);
        if (PRIMARY && from_keys) {
            // What face_raster_kernel left in the pixel: complement of (distance bits, inverted rank); zero = no face.
            // Then the faces too large for that kernel and the spheres, against this pixel's ray, under the same
            // rule: nearest wins, equal distances go to the later face, a sphere loses a tie (intersects(), :673-676).
            // A tile no face was accepted in holds no key (face_raster_kernel flags the tiles it touches); with the planes
            // prefilled and neither listed faces nor spheres to test, its pixels are done.
            int flagged = 1;
            if (ws.prefilled) { // (only such a pass sets flags)
                flagged = slot < total ? ws.tile_flag[slot >> 6] : 0;
                if (flagged && lane == 0) ws.tile_flag[slot >> 6] = 0; // read once per pass, left clean for the next one
                if (nbig_listed != 0 || sc.num_spheres != 0) flagged = 1;
            }
            if (valid && flagged) {
                float tbest = FLT_MAX;
                int best_rank = -1;
                unsigned long long *const cell = ws.pix_key + (row * fc.width + x); // (a frame's pixels fit 31 bits: frame_shape)
                const unsigned long long held = *cell;
                if (held != 0ull) {
                    *cell = 0ull; // every pixel's key is read exactly once per pass: left clean for the next one
                    tbest = __int_as_float((int)(unsigned)(~held >> 32));
                    best_rank = 0x7fffffff - (int)(unsigned)(~held & 0xffffffffull);
                    tri = sc.tri_of_rank[best_rank];
                }
                const int nbig = nbig_listed;
                if (nbig > 0 || sc.num_spheres > 0) {
                    const V3 o = mk(fc.origin[0], fc.origin[1], fc.origin[2]);
                    const V3 d = primary_direction(fc, ws, x, row);
                    if (KEEP) d_of[j] = d;
                    float ts = FLT_MAX;
                    if (sc.num_spheres > 0 && spheres_hit(sc, o, d, fc.t_eps, ts) && ts < tbest) tbest = ts, tri = -1, best_rank = -1;
                    const int gy = global_row(fc, row);
                    for (int e = 0; e < nbig; ++e) {
                        // (LDS or memory by a branch the whole workgroup takes alike: one pointer chosen between the two
                        // would turn every fetch into a flat access)
                        int4 g;
                        float4 pn;
                        if (big_in_lds)
                            g = s_big_box[e], pn = s_big_rec[e][0];
                        else
                            g = ws.big[e];
                        if (x < (g.y & 0xffff) || x > (g.z & 0xffff) || gy < (g.y >> 16) || gy > (g.z >> 16)) continue;
                        const float4 *rec = sc.tris + (size_t)g.x * 4;
                        if (!big_in_lds) pn = rec[0];
                        const V3 n = mk(pn.x, pn.y, pn.z);
                        const float t = (pn.w - dot(o, n)) / dot(d, n);
                        if (!(t >= fc.t_eps && t <= tbest)) continue; // also rejects NaN (see file header)
                        float4 r0, r1, r2;
                        if (big_in_lds)
                            r0 = s_big_rec[e][1], r1 = s_big_rec[e][2], r2 = s_big_rec[e][3];
                        else
                            r0 = rec[1], r1 = rec[2], r2 = rec[3];
                        const int rank = __float_as_int(r1.w);
                        if (t == tbest && rank < best_rank) continue; // equal distance: the later face wins (:760)
                        const V3 v0 = mk(r0.x, r0.y, r0.z), v1 = mk(r1.x, r1.y, r1.z), v2 = mk(r2.x, r2.y, r2.z);
                        const V3 hp = o + t * d;
                        if (dot(cross(v1 - v0, hp - v0), n) < 0) continue;
                        if (dot(cross(v2 - v1, hp - v1), n) < 0) continue;
                        if (dot(cross(v0 - v2, hp - v2), n) < 0) continue;
                        tbest = t, tri = g.x, best_rank = rank;
                    }
                }
                t_found = tri != -2 ? tbest : 0.0f;
            }
        } else {
            tri = valid ? ws.slot_tri[slot] : -2;
            t_found = valid && tri != -2 ? ws.slot_t[slot] : 0.0f;
            // the face raster ran but gave way to the tree walk: its keys and tile flags are dropped here
            if (PRIMARY && ws.pix_key != nullptr && valid) ws.pix_key[(size_t)row * (size_t)fc.width + (size_t)x] = 0ull;
            if (PRIMARY && ws.pix_key != nullptr && ws.prefilled && slot < total && lane == 0) ws.tile_flag[slot >> 6] = 0;
        }
        t_of[j] = t_found;
        const bool hit = tri != -2;
        const bool miss = list_misses && valid && !hit;
        // does this hit reflect?  src/flyscene.cpp:617-632 (a sphere ends the path, :618-621)
        bool go = false;
        if (spawn && hit && tri >= 0) {
            const int mat = __float_as_int(sc.shade[tri * kShadeRecordQuads + 5].y);
            if (mat >= 0 && mat < sc.num_materials) {
                const float *mp = sc.materials + kMaterialFloats * mat;
                const float k0 = mp[6], k1 = mp[7], k2 = mp[8]; // (fetched together: one round trip, not up to three)
                go = (k0 > fc.refl_thresh) | (k1 > fc.refl_thresh) | (k2 > fc.refl_thresh); // :631-632
            }
        }
        tri_of[j] = tri;
        hit_bits |= (unsigned)hit << j, go_bits |= (unsigned)go << j, miss_bits |= (unsigned)miss << j;
        const unsigned long long m_hit = __ballot(hit), m_go = __ballot(go), m_miss = __ballot(miss);
        if (lane == 0) {
            s_count[0][j][wave] = __popcll(m_hit);
            s_count[1][j][wave] = __popcll(m_go);
            s_count[2][j][wave] = __popcll(m_miss);
        }
    }
    __syncthreads();
    if (tid < 3) {
        int n = 0;
        for (int j = 0; j < kCompactSegments; ++j) n += s_count[tid][j][0] + s_count[tid][j][1] + s_count[tid][j][2] + s_count[tid][j][3];
        int *counter = tid == 0 ? &ws.stage[stage].hits : tid == 1 ? &ws.stage[stage].next : &ws.stage[stage].misses;
        s_base[tid] = n ? atomicAdd(counter, n) : 0;
    }
    __syncthreads();

    // ---- pass 2: write, segment by segment, wave by wave, lane by lane: slot order is kept inside the workgroup
    const unsigned long long below = (1ull << lane) - 1ull;
    int run_hit = s_base[0], run_go = s_base[1], run_miss = s_base[2];
#pragma unroll
    for (int j = 0; j < kCompactSegments; ++j) {
        const int slot = first + j * 256 + tid;
        const int tri = tri_of[j];
        const bool hit = (hit_bits >> j) & 1u, go = (go_bits >> j) & 1u, miss = (miss_bits >> j) & 1u;
        const unsigned long long m_hit = __ballot(hit), m_go = __ballot(go), m_miss = __ballot(miss);
        int pos_hit = run_hit, pos_go = run_go, pos_miss = run_miss;
        for (int w = 0; w < 4; ++w) {
            const int ch = s_count[0][j][w], cg = s_count[1][j][w], cm = s_count[2][j][w];
            if (w < wave) pos_hit += ch, pos_go += cg, pos_miss += cm;
            run_hit += ch, run_go += cg, run_miss += cm;
        }
        pos_hit += __popcll(m_hit & below), pos_go += __popcll(m_go & below), pos_miss += __popcll(m_miss & below);

        const float t = t_of[j];
        V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
        int pix = 0;
        if (PRIMARY) {
            int x = 0, row = 0;
            const bool valid = slot_pixel_of_wave(fc, slot, x, row) && slot < total;
            pix = row * fc.width + x;
            const bool filled = from_keys && ws.prefilled; // prep_kernel wrote what a miss leaves
            if (valid && (hit || !filled)) {
                if (face_id) face_id[pix] = tri >= 0 ? __float_as_int(sc.tris[tri * 4 + 1].w) : tri;
                if (t_hit) t_hit[pix] = t;
                if (!hit) { // background of a primary ray, :653-658
                    float *p = rgb + 3 * (size_t)pix;
                    if (fc.accumulate) {
                        p[0] += fc.bg[0], p[1] += fc.bg[1], p[2] += fc.bg[2];
                    } else {
                        p[0] = fc.bg[0], p[1] = fc.bg[1], p[2] = fc.bg[2];
                    }
                }
            }
            o = mk(fc.origin[0], fc.origin[1], fc.origin[2]);
            if (hit) d = have_d ? d_of[j] : primary_direction(fc, ws, x, row);
        } else if (hit) {
            const float *q = ws.queue[depth % 3];
            o = mk(plane(q, ws.capacity, 0)[slot], plane(q, ws.capacity, 1)[slot], plane(q, ws.capacity, 2)[slot]);
            d = mk(plane(q, ws.capacity, 3)[slot], plane(q, ws.capacity, 4)[slot], plane(q, ws.capacity, 5)[slot]);
            pix = __float_as_int(plane(q, ws.capacity, 9)[slot]);
        }
        const V3 hp = o + t * d; // hitPoint, :709,771
        if (go) {
            const float4 s4 = sc.shade[tri * kShadeRecordQuads + 4], s5 = sc.shade[tri * kShadeRecordQuads + 5];
            const V3 fn = mk(s4.z, s4.w, s5.x);
            const V3 refl = normalized(d - (2 * dot(d, fn)) * fn); // :785 then :635
            const V3 no = hp + fc.bounce_off * refl;               // :636
            float *q = ws.queue[(depth + 1) % 3];
            plane(q, ws.capacity, 0)[pos_go] = no.x, plane(q, ws.capacity, 1)[pos_go] = no.y, plane(q, ws.capacity, 2)[pos_go] = no.z;
            plane(q, ws.capacity, 3)[pos_go] = refl.x, plane(q, ws.capacity, 4)[pos_go] = refl.y, plane(q, ws.capacity, 5)[pos_go] = refl.z;
            plane(q, ws.capacity, 9)[pos_go] = __int_as_float(pix);
        }
        if (hit) {
            hl.slot[pos_hit] = slot;
            hl.tri[pos_hit] = tri;
            if (spawn) hl.child[pos_hit] = go ? pos_go : -1;
            plane(hl.p, ws.capacity, 0)[pos_hit] = hp.x;
            plane(hl.p, ws.capacity, 1)[pos_hit] = hp.y;
            plane(hl.p, ws.capacity, 2)[pos_hit] = hp.z;
        }
        if (miss) hl.slot[ws.capacity - 1 - pos_miss] = slot; // hits + misses <= rays <= capacity: the two ends never meet
    }
    __syncthreads(); // the counts in LDS are reused by the next run
    }
}

template <bool PRIMARY, int kCompactSegments, bool KEEP = false>
__global__ __launch_bounds__(256) void compact_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int stage, int depth, int spawn, int list_misses,
                                                      float *rgb, int32_t *face_id, float *t_hit, int total_static) {
    compact_body<PRIMARY, kCompactSegments, KEEP>(sc, fc, ws, stage, depth, spawn, list_misses, rgb, face_id, t_hit, total_static, (int)blockIdx.x, (int)gridDim.x);
}

// A bounce ray that hits nothing adds throughput * background_deep (zero in the reference,
// src/flyscene.cpp:661-665, so this kernel only runs when a caller sets that colour).
__global__ __launch_bounds__(256) void miss_kernel(FrameConst fc, Workspace ws, int stage, int depth, float *rgb) {
    const int misses = ws.stage[stage].misses;
    for (int idx = (int)(blockIdx.x * 256 + threadIdx.x); idx < misses; idx += (int)gridDim.x * 256) {
        const int slot = ws.hits[depth & 1].slot[ws.capacity - 1 - idx];
        const float *q = ws.queue[depth % 3];
        float *p = rgb + 3 * (size_t)__float_as_int(plane(q, ws.capacity, 9)[slot]);
        p[0] += plane(q, ws.capacity, 6)[slot] * fc.bg_deep[0];
        p[1] += plane(q, ws.capacity, 7)[slot] * fc.bg_deep[1];
        p[2] += plane(q, ws.capacity, 8)[slot] * fc.bg_deep[2];
    }
}

// ---------------------------------------------------------------------------------------------
// Shading: one lane per entry of the hit list.
// ---------------------------------------------------------------------------------------------
template <bool PRIMARY>
__device__ __forceinline__ void shade_body(const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, int spawn, float *rgb, int block,
                                           int nblocks) {
    const int nhits = ws.stage[stage].hits;
    // Report what this stage held to the host (page-locked memory, never read by the device): the next frames of the
    // view size their launches from it instead of from the worst case.  Stored + 1: zero means "nothing reported yet".
    if (block == 0 && threadIdx.x == 0 && ws.stage_feedback) {
        ws.stage_feedback[2 * stage] = nhits + 1;
        ws.stage_feedback[2 * stage + 1] = ws.stage[stage].next + 1;
    }
    const HitList hl = ws.hits[depth & 1];
    // the lights' positions once per workgroup (read per hit from global memory they are a fetch per light, one after the other)
    constexpr int kLdsLights = 64;
    __shared__ float s_light[3 * kLdsLights];
    const bool lights_in_lds = sc.num_shade_lights <= kLdsLights;
    if (lights_in_lds) {
        for (int i = (int)threadIdx.x; i < 3 * sc.num_shade_lights; i += 256) s_light[i] = sc.light_pos[i];
        __syncthreads();
    }
    for (int idx = block * 256 + (int)threadIdx.x; idx < nhits; idx += nblocks * 256) {

    const V3 P = mk(plane(hl.p, ws.capacity, 0)[idx], plane(hl.p, ws.capacity, 1)[idx], plane(hl.p, ws.capacity, 2)[idx]);
    // getLightIntensity, src/flyscene.cpp:870-896: share of unoccluded samples per light, summed, capped at 1
    // (the hit's other fields are asked for here, with the first samples: everything that depends on the index alone in
    // one round trip; the samples four at a time, not one fetch after the other -- they are 0 or 1, so any order sums alike)
    const int slot = hl.slot[idx];
    const int tri = hl.tri[idx];
    const int child = spawn ? hl.child[idx] : -1;
    float li = 0.0f;
    if (sc.spl >= 4) {
        for (int l = 0; l < sc.num_lights; ++l) {
            const uint8_t *const seen = ws.clear[depth & 1] + (size_t)(l * sc.spl) * ws.capacity + idx;
            float reaching = 0.0f;
            int k = 0;
            for (; k + 4 <= sc.spl; k += 4) {
                const uint8_t b0 = seen[(size_t)k * ws.capacity], b1 = seen[(size_t)(k + 1) * ws.capacity];
                const uint8_t b2 = seen[(size_t)(k + 2) * ws.capacity], b3 = seen[(size_t)(k + 3) * ws.capacity];
                reaching += (float)b0, reaching += (float)b1, reaching += (float)b2, reaching += (float)b3;
            }
            for (; k < sc.spl; ++k) reaching += (float)seen[(size_t)k * ws.capacity];
            li += reaching / (float)sc.spl;
        }
    } else {
        // few samples per light (hard shadows of many lights): four fetches at a time across the lights, summed light by
        // light in the same order
        const uint8_t *const seen = ws.clear[depth & 1] + idx;
        const int nshadow = sc.num_lights * sc.spl;
        float reaching = 0.0f;
        int of_light = 0;
        for (int m0 = 0; m0 < nshadow; m0 += 4) {
            uint8_t b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = m0 + u < nshadow ? seen[(size_t)(m0 + u) * ws.capacity] : (uint8_t)0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (m0 + u >= nshadow) break;
                reaching += (float)b[u];
                if (++of_light == sc.spl) {
                    li += reaching / (float)sc.spl;
                    reaching = 0.0f, of_light = 0;
                }
            }
        }
    }
    li = smin(li, 1.0f);
    const V3 LI = mk(li, li, li);

    // the ray that produced this hit
    V3 ro, T, ks = mk(0, 0, 0);
    int pix;
    if (PRIMARY) {
        int x, row;
        slot_pixel(fc, slot, x, row);
        pix = row * fc.width + x;
        ro = mk(fc.origin[0], fc.origin[1], fc.origin[2]);
        T = mk(1.0f, 1.0f, 1.0f);
    } else {
        const float *q = ws.queue[depth % 3];
        ro = mk(plane(q, ws.capacity, 0)[slot], plane(q, ws.capacity, 1)[slot], plane(q, ws.capacity, 2)[slot]);
        T = mk(plane(q, ws.capacity, 6)[slot], plane(q, ws.capacity, 7)[slot], plane(q, ws.capacity, 8)[slot]);
        pix = __float_as_int(plane(q, ws.capacity, 9)[slot]);
    }

    // shadeOffFace, src/flyscene.cpp:461-579
    V3 local;
    if (tri < 0) {
        local = mk(0.7f, 0.9f, 0.0f); // sphere, :463-465
    } else {
        float4 s0 = sc.shade[tri * kShadeRecordQuads + 0], s1 = sc.shade[tri * kShadeRecordQuads + 1], s2 = sc.shade[tri * kShadeRecordQuads + 2];
        float4 s3 = sc.shade[tri * kShadeRecordQuads + 3], s4 = sc.shade[tri * kShadeRecordQuads + 4], s5 = sc.shade[tri * kShadeRecordQuads + 5];
        keep(s0), keep(s1), keep(s2), keep(s3), keep(s4), keep(s5); // (the whole record at once, not the material index first)
        const V3 v0 = mk(s0.x, s0.y, s0.z), v1 = mk(s0.w, s1.x, s1.y), v2 = mk(s1.z, s1.w, s2.x);
        const V3 n0 = mk(s2.y, s2.z, s2.w), n1 = mk(s3.x, s3.y, s3.z), n2 = mk(s3.w, s4.x, s4.y);
        const int mat = __float_as_int(s5.y);
        if (mat < 0 || mat >= sc.num_materials) {
            local = mk(0.5f, 0.5f, 0.5f); // :469-472
        } else {
            const float *mp = sc.materials + kMaterialFloats * mat;
            const V3 ka = mk(mp[0], mp[1], mp[2]), kd = mk(mp[3], mp[4], mp[5]);
            ks = mk(mp[6], mp[7], mp[8]);
            const float shininess = mp[9];
            // area-weighted normal, :494-500
            const float a12 = norm(cross(v1 - P, v2 - P)) * 0.5f;
            const float a02 = norm(cross(v0 - P, v2 - P)) * 0.5f;
            const float a01 = norm(cross(v0 - P, v1 - P)) * 0.5f;
            const V3 N = normalized((a12 * n0 + a02 * n1) + a01 * n2);
            // Texture branch, :476-481,516-542.  The weights are the absolute sub-triangle areas, not divided
            // by the face area; uacc and vacc of the reference are floors of fractional parts, always 0, so
            // its "interpolation" is the one texel at (projU, projV).  The reference indexes unchecked; an
            // index outside the image (undefined there) is clamped to it.
            V3 albedo = kd;
            const int tex = sc.num_textures > 0 ? __float_as_int(mp[10]) : -1;
            if (tex >= 0) {
                const float4 s6 = sc.shade[tri * kShadeRecordQuads + 6];
                const float u = (a12 * s5.z + a02 * s6.x) + a01 * s6.z;
                const float v = (a12 * s5.w + a02 * s6.y) + a01 * s6.w;
                const int first = sc.tex_info[4 * tex], w = sc.tex_info[4 * tex + 1], h = sc.tex_info[4 * tex + 2];
                const double texel = (double)floorf(v * (float)h) * (double)w + (double)floorf(u * (float)w);
                const double last = (double)w * (double)h - 1.0;
                const int at = !(texel >= 0.0) ? 0 : texel > last ? (int)last : (int)texel;
                const float *tx = sc.texels + 3 * (size_t)(first + at);
                albedo = mk(tx[0], tx[1], tx[2]);
            }
            local = mk(0.0f, 0.0f, 0.0f);
            for (int l = 0; l < sc.num_shade_lights; ++l) { // every entry of `lights` (:505), sampled or not
                const V3 lp = lights_in_lds ? mk(s_light[3 * l], s_light[3 * l + 1], s_light[3 * l + 2])
                                            : mk(sc.light_pos[3 * l], sc.light_pos[3 * l + 1], sc.light_pos[3 * l + 2]);
                const V3 L = normalized(lp - P);                          // :507
                const V3 ambient = cmul(LI, ka);                          // :510
                const float cos1 = fmaxf(0.0f, dot(L, N));                // :513
                const V3 diffuse = cos1 * cmul(LI, albedo);               // :514 / :541
                const V3 E = normalized(ro - P);                          // :545
                const V3 R = normalized(-L + (2.f * dot(L, N)) * N);      // :546
                const float cos2 = fmaxf(0.0f, dot(R, E));                // :548
                const V3 specular = powf(cos2, shininess) * cmul(LI, ks); // :549
                const V3 sum = (diffuse + specular) + ambient;            // :554
                local = local + mk(smin(smax(sum.x, 0.0f), 1.0f), smin(smax(sum.y, 0.0f), 1.0f), smin(smax(sum.z, 0.0f), 1.0f));
            }
        }
    }

    // pixel += throughput * local.  Each pixel owns at most one live path, so no atomics are needed.
    float *p = rgb + 3 * (size_t)pix;
    if (PRIMARY && !fc.accumulate) {
        p[0] = local.x, p[1] = local.y, p[2] = local.z;
    } else {
        const V3 add = cmul(T, local);
        p[0] += add.x, p[1] += add.y, p[2] += add.z;
    }

    // throughput of the reflection ray spawned from this hit: local + C_next * ks * LI, :637,645
    if (spawn) {
        if (child >= 0) {
            const V3 nT = cmul(cmul(T, ks), LI);
            float *q = ws.queue[(depth + 1) % 3];
            plane(q, ws.capacity, 6)[child] = nT.x, plane(q, ws.capacity, 7)[child] = nT.y, plane(q, ws.capacity, 8)[child] = nT.z;
        }
    }
    }
}

template <bool PRIMARY>
__global__ __launch_bounds__(256) void shade_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int stage, int depth, int spawn, float *rgb) {
    shade_body<PRIMARY>(sc, fc, ws, stage, depth, spawn, rgb, (int)blockIdx.x, (int)gridDim.x);
}

// Shading of recursion level `depth` and compaction of the closest hits of level depth + 1 in one launch: the two are
// independent (the compaction reads the geometry of the reflection rays, the shading writes their throughput), and each
// is short enough that what a launch costs beside its work shows.
template <bool PRIMARY>
__global__ __launch_bounds__(256) void shade_compact_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int stage, int depth, int spawn_next,
                                                            int list_misses, float *rgb, int compact_blocks) {
    if ((int)blockIdx.x < compact_blocks)
        compact_body<false, 2>(sc, fc, ws, stage + 1, depth + 1, spawn_next, list_misses, rgb, nullptr, nullptr, 0, (int)blockIdx.x, compact_blocks);
    else
        shade_body<PRIMARY>(sc, fc, ws, stage, depth, /*spawn=*/1, rgb, (int)blockIdx.x - compact_blocks, (int)gridDim.x - compact_blocks);
}

// Start of a pass: zero the counters the pass counts up from (`zero16` 16-byte words from `zero`), and fill the
// tables of the pass's screen coordinates (primary_direction): dir_x per image column, dir_y per LOCAL row.
// With ws.prefilled it also writes what a primary ray that hits nothing leaves in the output planes (background colour,
// RTAMD_FACE_MISS, distance 0: src/flyscene.cpp:650-658) into EVERY pixel, in whole coalesced lines; the compaction
// kernel then skips the tiles no face was accepted in, and the shading overwrites the pixels that were hit.
__global__ __launch_bounds__(256) void prep_kernel(FrameConst fc, Workspace ws, int4 *zero, int zero16, int4 *zero_b, int zero_b16, float *rgb,
                                                   int32_t *face_id, float *t_hit) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (ws.prefilled && !ws.prefill_with_faces) prefill_planes(fc, rgb, face_id, t_hit, i, (int)gridDim.x * 256);
    for (int k = i; k < zero16; k += (int)gridDim.x * 256) zero[k] = make_int4(0, 0, 0, 0);
    for (int k = i; k < zero_b16; k += (int)gridDim.x * 256) zero_b[k] = make_int4(0, 0, 0, 0);
    const int nx = fc.tiles_x * 8, ny = fc.tiles_y * 8;
    for (int k = i; k < nx + ny; k += (int)gridDim.x * 256) {
        if (k < nx)
            ws.dir_x[k] = screen_x(fc, (float)k + fc.sub_x);
        else
            ws.dir_y[k - nx] = screen_y(fc, (float)global_row(fc, k - nx) + fc.sub_y);
    }
}

// pixel /= SSAALEVEL^2, src/flyscene.cpp:830
__global__ void scale_kernel(float *rgb, size_t n, float divisor) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rgb[i] = rgb[i] / divisor;
}

// The bytes writePPMImage puts in the file: min(255, (int)(255 * c)) per channel (utils/ppmIO.hpp:203).  Four
// channels per thread, one 4-byte store; n4 = number of whole groups of four, the tail goes one by one.
__device__ __forceinline__ unsigned ppm_byte(float c) {
    const float v = 255.0f * c;
    // (int) of a value outside the int range is undefined in C; the device conversion saturates, NaN gives 0
    const int i = (int)v;
    return (unsigned)max(0, min(255, i)); // the host writer clamps below as well
}
__global__ void quantize_kernel(const float *rgb, uint8_t *out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const float4 v = *(const float4 *)(rgb + i);
        *(unsigned *)(out + i) = ppm_byte(v.x) | (ppm_byte(v.y) << 8) | (ppm_byte(v.z) << 16) | (ppm_byte(v.w) << 24);
    } else {
        for (size_t k = i; k < n; ++k) out[k] = (uint8_t)ppm_byte(rgb[k]);
    }
}

// The same bytes, and beside them whether they say all there is to say: writePPMImage prints min(255, (int)(255 * c)), which
// is negative for c <= -1/255 and undefined for a NaN or a value beyond the int range; a byte cannot hold either.  Any such
// channel sets *inexact (an int the caller cleared), and the caller falls back to the float framebuffer.
__global__ void quantize_checked_kernel(const float *rgb, uint8_t *out, size_t n, int *inexact) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    bool odd = false;
    if (i + 4 <= n) {
        const float4 v = *(const float4 *)(rgb + i);
        *(unsigned *)(out + i) = ppm_byte(v.x) | (ppm_byte(v.y) << 8) | (ppm_byte(v.z) << 16) | (ppm_byte(v.w) << 24);
        const float lo = fminf(fminf(v.x, v.y), fminf(v.z, v.w)), hi = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
        odd = !(255.0f * lo > -1.0f) || !(255.0f * hi < 2147483648.0f) || v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w;
    } else {
        for (size_t k = i; k < n; ++k) {
            out[k] = (uint8_t)ppm_byte(rgb[k]);
            odd = odd || !(255.0f * rgb[k] > -1.0f) || !(255.0f * rgb[k] < 2147483648.0f);
        }
    }
    if (__ballot(odd) != 0ull && (threadIdx.x & 63) == 0) *inexact = 1; // (plain store: every writer stores the same value)
}

// Packed part-major rows -> image rows (4-byte elements).
__global__ void assemble_kernel(const uint32_t *src, uint32_t *dst, int width, int height, int channels, int row_tile, int num_parts) {
    const size_t row_elems = (size_t)width * channels;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_elems * (size_t)height) return;
    const int y = (int)(i / row_elems);
    const size_t x = i - (size_t)y * row_elems;
    const int tile = y / row_tile, part = tile % num_parts, ltile = tile / num_parts;
    // rows owned by parts before `part`
    const int ntiles = (height + row_tile - 1) / row_tile;
    size_t rows_before = 0;
    for (int p = 0; p < part; ++p) {
        const int tcount = (ntiles - p + num_parts - 1) / num_parts; // tiles p, p+P, ...
        size_t rows = (size_t)tcount * row_tile;
        const int last_tile = p + (tcount - 1) * num_parts;
        if (tcount > 0 && (last_tile + 1) * row_tile > height) rows -= (size_t)((last_tile + 1) * row_tile - height);
        rows_before += rows;
    }
    const size_t local_row = (size_t)ltile * row_tile + (size_t)(y - tile * row_tile);
    dst[i] = src[(rows_before + local_row) * row_elems + x];
}

} // namespace

// ---------------------------------------------------------------------------------------------
// Launchers (host).  Shapes are validated by the caller (api.cpp) before any launch.
// ---------------------------------------------------------------------------------------------
namespace {
// Worst-case grid of one-wave workgroups, capped (surplus groups are then dealt round-robin) and kept a
// multiple of 8 so that a wave's XCD family does not change between its round-robin trips.  The grid must
// cover the worst case because the ray count is only known on the device; most of it usually finds nothing
// to do, and launching an idle workgroup is not free: 421 K of them (1080p x 13 samples) cost about 90 us,
// as much as the shadow rays of the frame's 160 K hits themselves.  2^17 keeps sixteen rounds of work per
// wave slot for the balance of large launches.
int walk_grid(long long groups) {
    static const long long cap = [] { // experiment hook
        const char *e = getenv("RTAMD_WALK_MAX_BLOCKS");
        const long long n = e ? atoll(e) : 0;
        return n >= 8 ? n : (1ll << 17);
    }();
    long long b = groups < cap ? groups : cap;
    b = (b + 7) & ~7ll;
    return (int)(b < 8 ? 8 : b);
}
} // namespace

float raster_margin_floor() { return kBinMarginFloor; }

void launch_prep(const FrameConst &fc, const Workspace &ws, void *zero, size_t zero_bytes, void *zero_b, size_t zero_b_bytes, float *rgb,
                 int32_t *face_id, float *t_hit, hipStream_t stream) {
    const int zero16 = (int)((zero_bytes + 15) / 16), zero_b16 = (int)((zero_b_bytes + 15) / 16);
    int items = std::max(std::max(zero16, zero_b16), (fc.tiles_x + fc.tiles_y) * 8);
    const bool fills = ws.prefilled && !ws.prefill_with_faces;
    if (fills) items = std::max(items, (int)std::min<long long>(3ll * fc.local_rows * fc.width / 4, 1 << 30)); // four stores per thread
    const int blocks = std::max(1, std::min(fills ? 16384 : 2048, (items + 255) / 256));
    prep_kernel<<<blocks, 256, 0, stream>>>(fc, ws, (int4 *)zero, zero16, (int4 *)zero_b, zero_b16, rgb, face_id, t_hit);
}

// Primary rays of one pass face by face (ws.pix_key must be set and zeroed): one launch.  The pixels are resolved by
// launch_compact; the tree walk behind it (launch_walk_closest) only does anything where the pass gave way to it.
void launch_primary_faces(bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int num_tris, float *rgb, int32_t *face_id,
                          float *t_hit, hipStream_t stream) {
    if (num_tris <= 0 && !(ws.prefilled && ws.prefill_with_faces)) return;
    // Faces per wave: 64, or fewer while that leaves fewer than 4096 waves (the chip holds about 5000 of this kernel): down to 16 for
    // a frame alone on the GPU, whose few waves would each step through hundreds of pairs one batch after the other; down to 32
    // with frames in flight, which fill the chip between them and pay for every wave's projection with its idle lanes.
    // 69 K faces, frames in flight, ms per frame with 16 / 32 / 64 faces per wave: 0.0940 / 0.0922 / 0.0920 at 1080p, 0.0886 /
    // 0.0875 / 0.0869 at 4K; a frame alone: 0.179 / 0.181 / 0.187 and 0.142 / 0.156 / 0.184.
    static const int forced = getenv("RTAMD_RASTER_FACES_PER_WAVE_SHIFT") ? atoi(getenv("RTAMD_RASTER_FACES_PER_WAVE_SHIFT")) : -1; // experiment hook
    int fshift = 6;
    const int fewest = ws.full_waves ? 5 : 4;
    while (fshift > fewest && ((long long)num_tris >> fshift) < 4096) --fshift;
    if (forced >= 2 && forced <= 6) fshift = forced;
    const unsigned threads = 64u * (unsigned)kFilterWaves, per_block = (unsigned)kFilterWaves << fshift;
    const unsigned blocks = (unsigned)(((unsigned)std::max(num_tris, 0) + per_block - 1) / per_block);
    unsigned fill_blocks = 0; // (four stores per thread and trip, as in launch_prep)
    if (ws.prefilled && ws.prefill_with_faces)
        fill_blocks = (unsigned)std::max(1ll, std::min(4194304ll / threads, (3ll * fc.local_rows * fc.width / 4 + threads - 1) / threads));
    if (count)
        face_raster_kernel<true><<<blocks + fill_blocks, threads, 0, stream>>>(sc, fc, ws, num_tris, fshift, (int)blocks, rgb, face_id, t_hit);
    else
        face_raster_kernel<false><<<blocks + fill_blocks, threads, 0, stream>>>(sc, fc, ws, num_tris, fshift, (int)blocks, rgb, face_id, t_hit);
}

void launch_primary_bins(bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int num_tris, bool cleared,
                         hipStream_t stream) {
    const int total_static = fc.tiles_x * fc.tiles_y * 64;
    // bin_state (256 bytes) and bin_count are neighbours; the caller clears them with the frame's counters for the first pass
    if (!cleared) (void)hipMemsetAsync(ws.bin_state, 0, 256 + sizeof(int32_t) * (size_t)ws.num_bins, stream);
    // Small faces (less than ten pixels of the whole image each, on average) share their atomics wave by wave.  Not
    // in a partitioned frame: a rank bins only the rows it owns, the atomics are few, and the shared loop only costs
    // (one rank of eight at 4K: 0.188 vs 0.201 ms).
    static const int forced = getenv("RTAMD_BIN_AGGREGATE") ? atoi(getenv("RTAMD_BIN_AGGREGATE")) : -1; // experiment hook
    const bool aggregate = forced >= 0 ? forced != 0 : fc.num_parts == 1 && 10ll * num_tris > (long long)fc.width * fc.height;
    const unsigned face_blocks = (unsigned)((num_tris + 255) / 256);
    if (num_tris > 0) {
        if (aggregate)
            bin_count_kernel<true><<<face_blocks, 256, 0, stream>>>(sc, fc, ws, num_tris);
        else
            bin_count_kernel<false><<<face_blocks, 256, 0, stream>>>(sc, fc, ws, num_tris);
    }
    bin_offsets_kernel<<<(ws.num_bins + 256 * kOffsetItems - 1) / (256 * kOffsetItems), 256, 0, stream>>>(ws);
    if (num_tris > 0) {
        if (aggregate)
            bin_fill_kernel<true><<<face_blocks, 256, 0, stream>>>(sc, fc, ws, num_tris);
        else
            bin_fill_kernel<false><<<face_blocks, 256, 0, stream>>>(sc, fc, ws, num_tris);
    }
    if (count)
        raster_kernel<true><<<(total_static + 255) / 256, 256, 0, stream>>>(sc, fc, ws, total_static);
    else
        raster_kernel<false><<<(total_static + 255) / 256, 256, 0, stream>>>(sc, fc, ws, total_static);
}

// `expect`: what an earlier frame of the same view held at this point (rays of a bounce stage, hits of a stage), or
// -1 when nothing is known.  Launches whose size is only known on the device cover the worst case then; with a figure
// they are cut to it plus a quarter.  Every kernel sized this way strides over its work, so a figure that turns out
// too small costs time, never results.
namespace {
long long with_margin(int expect) { return (long long)expect + (expect >> 2) + 64; }
} // namespace

// Grid of the primary tree walk behind a screen-space pass that earlier frames of the view saw accepted: the launch then only
// finds out that it has nothing to do (it strides over the pixels should the pass have given way after all).
#ifndef RT_FALLBACK_WALK_BLOCKS
#define RT_FALLBACK_WALK_BLOCKS 256
#endif
void launch_walk_closest(bool primary, bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth,
                         int expect, hipStream_t stream) {
    const int total_static = fc.tiles_x * fc.tiles_y * 64;
    // bounce queues are at most as long as the slot count: that is the worst case the grid covers
    // With screen binning in use the primary walk only runs when the pass turned its bins down (overflow, or
    // lists too long to pay): a moderate grid (the kernel strides over the groups) keeps the usual early exit cheap.
    long long groups = primary ? fc.tiles_x * fc.tiles_y : ws.capacity / 64;
    if (!primary && expect >= 0) // narrow groups (a launch alone on the chip) hold as few as sixteen rays
        groups = std::min(groups, (with_margin(expect) + (ws.full_waves ? 63 : (1 << kMinGroupShift) - 1)) / (ws.full_waves ? 64 : 1 << kMinGroupShift));
    const int blocks = primary && (ws.bin_entries || ws.pix_key) ? (ws.bins_expected ? RT_FALLBACK_WALK_BLOCKS : 16384) : walk_grid(groups);
    static const int steal = getenv("RTAMD_WALK_STEAL") ? atoi(getenv("RTAMD_WALK_STEAL")) : 1; // experiment hook
    if (steal && fc.t_eps > 0.0f) { // (the shared nearest distance is ordered by its bits: distances must be positive, -0.0f included)
        if (primary) {
            if (count)
                walk_steal_kernel<kPrimary, true><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, total_static);
            else
                walk_steal_kernel<kPrimary, false><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, total_static);
        } else {
            if (count)
                walk_steal_kernel<kQueue, true><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, 0);
            else
                walk_steal_kernel<kQueue, false><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, 0);
        }
        return;
    }
    if (primary) {
        if (count)
            walk_kernel<kPrimary, true><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, total_static);
        else
            walk_kernel<kPrimary, false><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, total_static);
    } else {
        if (count)
            walk_kernel<kQueue, true><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, 0);
        else
            walk_kernel<kQueue, false><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, 0);
    }
}

void launch_compact(bool primary, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, bool spawn,
                    bool list_misses, float *rgb, int32_t *face_id, float *t_hit, int expect, hipStream_t stream, int expect_hits) {
    const int total_static = fc.tiles_x * fc.tiles_y * 64;
    long long slots = primary ? total_static : ws.capacity;
    if (!primary && expect >= 0) slots = std::min(slots, with_margin(expect));
    // Slots per workgroup (one atomic per list and workgroup): 512 for small images, which need the workgroups to fill the chip;
    // 2048 for large ones where most workgroups only step over tiles nothing was hit in; 1024 for large ones with hits in a
    // quarter of the pixels or more (what earlier frames of the view reported: `expect_hits`), where a thread's segments,
    // taken one after the other, are a long chain of fetches (1.1 M-face field at 4K: this kernel 0.144 -> 0.118 ms).
    const bool large = slots > (4 << 20);
    const bool dense = primary && expect_hits >= 0 && 4ll * expect_hits >= slots;
    const int segments = !large ? 2 : dense ? 4 : 8;
    const int per_block = 256 * segments;
    const int blocks = (int)((slots + per_block - 1) / per_block);
    if (primary && ws.listed_faces_expected && segments == 4) { // (1.1 M-face field on a ground plane at 4K: this kernel 102 -> 96 us)
        compact_kernel<true, 4, true><<<blocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn, 0, rgb, face_id, t_hit, total_static);
    } else if (primary && ws.listed_faces_expected && segments == 8) {
        compact_kernel<true, 8, true><<<blocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn, 0, rgb, face_id, t_hit, total_static);
    } else if (primary) {
        if (segments == 8)
            compact_kernel<true, 8><<<blocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn, 0, rgb, face_id, t_hit, total_static);
        else if (segments == 4)
            compact_kernel<true, 4><<<blocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn, 0, rgb, face_id, t_hit, total_static);
        else
            compact_kernel<true, 2><<<blocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn, 0, rgb, face_id, t_hit, total_static);
    } else {
        if (large)
            compact_kernel<false, 8><<<blocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn, list_misses, rgb, face_id, t_hit, 0);
        else
            compact_kernel<false, 2><<<blocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn, list_misses, rgb, face_id, t_hit, 0);
    }
}

void launch_walk_shadow(bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, int expect,
                        hipStream_t stream) {
    const int nshadow = sc.num_lights * sc.spl;
    if (nshadow <= 0) return;
    long long groups = (long long)(fc.tiles_x * fc.tiles_y) * nshadow;
    if (expect >= 0) { // small launches put sixteen lanes on a ray: four hits per group
        const long long hits = with_margin(expect);
        const bool split = sc.use_grids && hits * nshadow < RT_SHADOW_SPLIT_BELOW;
        groups = std::min(groups, (split ? (hits + 3) / 4 : (hits + 63) / 64) * nshadow);
    }
    const int blocks = sc.use_grids ? (walk_grid(groups) + 7) & ~7 : walk_grid(groups); // (whole rounds over the XCDs: shadow_grid_body)
    if (sc.use_grids) {
        if (count)
            shadow_grid_kernel<true><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth);
        else
            shadow_grid_kernel<false><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth);
        return;
    }
    if (count)
        walk_kernel<kShadow, true><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, 0);
    else
        walk_kernel<kShadow, false><<<blocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, 0);
}

void launch_shade(bool primary, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, bool spawn, float *rgb,
                  int expect, hipStream_t stream) {
    long long hits = ws.capacity;
    if (expect >= 0) hits = std::min(hits, with_margin(expect));
    const int sblocks = (int)((hits + 255) / 256);
    if (primary)
        shade_kernel<true><<<sblocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn, rgb);
    else
        shade_kernel<false><<<sblocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn, rgb);
}

namespace {
int bounce_walk_blocks(const Workspace &ws, int expect) {
    long long groups = ws.capacity / 64;
    if (expect >= 0) groups = std::min(groups, (with_margin(expect) + (ws.full_waves ? 63 : (1 << kMinGroupShift) - 1)) / (ws.full_waves ? 64 : 1 << kMinGroupShift));
    return walk_grid(groups);
}
int shadow_blocks(const DeviceScene &sc, const FrameConst &fc, int expect) {
    const int nshadow = sc.num_lights * sc.spl;
    long long groups = (long long)(fc.tiles_x * fc.tiles_y) * nshadow;
    if (expect >= 0) { // small launches put sixteen lanes on a ray: four hits per group
        const long long hits = with_margin(expect);
        const bool split = sc.use_grids && hits * nshadow < RT_SHADOW_SPLIT_BELOW;
        groups = std::min(groups, (split ? (hits + 3) / 4 : (hits + 63) / 64) * nshadow);
    }
    return walk_grid(groups);
}
} // namespace

bool can_fuse_shadow_walk(const DeviceScene &sc, const FrameConst &fc) { return sc.use_grids && sc.num_lights * sc.spl > 0 && fc.t_eps > 0.0f; }
bool can_fuse_shade_compact(const Workspace &ws) { return ws.capacity <= (4 << 20); }

void launch_shadow_walk(bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, int expect_hits,
                        int expect_next, hipStream_t stream) {
    // both shares in whole rounds over the eight XCDs, so that a shadow workgroup's index within its share tells its XCD class
    const int wblocks = (bounce_walk_blocks(ws, expect_next) + 7) & ~7, sblocks = (shadow_blocks(sc, fc, expect_hits) + 7) & ~7;
    if (count)
        shadow_walk_kernel<true><<<wblocks + sblocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, wblocks);
    else
        shadow_walk_kernel<false><<<wblocks + sblocks, 64, 0, stream>>>(sc, fc, ws, stage, depth, wblocks);
}

void launch_shade_compact(bool primary, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int depth, bool spawn_next,
                          bool list_misses, float *rgb, int expect_hits, int expect_next, hipStream_t stream) {
    long long slots = ws.capacity, hits = ws.capacity;
    if (expect_next >= 0) slots = std::min(slots, with_margin(expect_next));
    if (expect_hits >= 0) hits = std::min(hits, with_margin(expect_hits));
    const int cblocks = (int)((slots + 511) / 512), sblocks = (int)((hits + 255) / 256);
    if (primary)
        shade_compact_kernel<true><<<cblocks + sblocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn_next, list_misses, rgb, cblocks);
    else
        shade_compact_kernel<false><<<cblocks + sblocks, 256, 0, stream>>>(sc, fc, ws, stage, depth, spawn_next, list_misses, rgb, cblocks);
}

void launch_misses(const FrameConst &fc, const Workspace &ws, int stage, int depth, float *rgb, int expect, hipStream_t stream) {
    long long rays = ws.capacity;
    if (expect >= 0) rays = std::min(rays, with_margin(expect));
    miss_kernel<<<(unsigned)((rays + 255) / 256), 256, 0, stream>>>(fc, ws, stage, depth, rgb);
}

void launch_scale(float *rgb, size_t n, float divisor, hipStream_t stream) {
    if (n == 0) return;
    scale_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(rgb, n, divisor);
}

void launch_quantize(const float *rgb, uint8_t *out, size_t n, hipStream_t stream) {
    if (n == 0) return;
    const size_t threads = (n + 3) / 4;
    quantize_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(rgb, out, n);
}

void launch_quantize_checked(const float *rgb, uint8_t *out, size_t n, int *inexact, hipStream_t stream) {
    if (n == 0) return;
    const size_t threads = (n + 3) / 4;
    quantize_checked_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(rgb, out, n, inexact);
}

void launch_assemble(const void *src, void *dst, int width, int height, int channels, int row_tile, int num_parts, hipStream_t stream) {
    const size_t n = (size_t)width * height * channels;
    if (n == 0) return;
    assemble_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>((const uint32_t *)src, (uint32_t *)dst, width, height, channels, row_tile,
                                                                      num_parts);
}

} // namespace rtamd
