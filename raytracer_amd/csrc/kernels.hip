// gfx950 kernels of the Flyscene render pass.
//
// Two kernels per recursion level replace Flyscene::traceRay and everything below it:
//
//   trace_kernel  -- closest hit of one ray per lane: intersects() = sphereIntersection +
//                    triangleIntersection (src/flyscene.cpp:669-793), with boundingBox::
//                    intersectingBoxes (src/boundingBox.h:295-307) replaced by an ordered,
//                    distance-pruned walk of the flat BVH.  Level 0 also generates the camera ray
//                    (tracePixels, src/flyscene.cpp:808-822; Camera::screenToWorld, camera.hpp:158-176).
//   shade_kernel  -- per 64 hits: the shadow rays of getLightIntensity (src/flyscene.cpp:870-896) as
//                    any-hit queries spread over the workgroup's waves, then shadeOffFace
//                    (:461-579) and the reflection set-up of traceRay (:617-646) for the next level.
//
// Arithmetic contract: this file is compiled with -ffp-contract=off and without fast-math.  Every
// expression that decides a hit, a distance, a hit point or a shadow is written in the reference's
// evaluation order (Eigen's 3-vector reductions associate as x + (y + z)), so those values are
// bit-identical to a strict binary32 evaluation of the reference.  The slab tests are the only
// place where explicit fused multiply-adds and a reciprocal are used; they are conservative
// (the exit distance is widened), and a widened test can only add faces to be tested exactly.
// Deliberate deviation: a NaN plane distance (ray inside a face plane, 0/0) is rejected, whereas
// the reference accepts it and then returns a visiting-order-dependent face.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "device_types.h"
#include "kernels.h"

namespace rtamd {

namespace {

constexpr int kTraceThreads = 256;
constexpr int kSentinel = 0x7fffffff;

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

struct Ray {
    V3 o, d;
};

struct Work {
    unsigned box, plane, edge;
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---------------------------------------------------------------------------------------------
// Camera ray of one pixel sample: Camera::screenToWorld (camera.hpp:158-176, double intermediates
// kept) followed by (p - origin).normalized() (src/flyscene.cpp:822).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 camera_direction(const FrameConst &fc, float px, float py) {
    float nx = (float)(2.0 * (double)(px - fc.vp[0]) / (double)fc.vp[2] - 1.0);
    float ny = (float)(1.0 - 2.0 * (double)(py - fc.vp[1]) / (double)fc.vp[3]);
    const float nz = -1.0f;
    nx *= fc.ax;
    ny *= fc.scale;
    const float *m = fc.inv_view;
    const V3 w = mk(m[3] + (m[0] * nx + m[1] * ny + m[2] * nz), m[7] + (m[4] * nx + m[5] * ny + m[6] * nz),
                    m[11] + (m[8] * nx + m[9] * ny + m[10] * nz));
    return normalized(w - mk(fc.origin[0], fc.origin[1], fc.origin[2]));
}

// Image row of a local (packed) row under the row-tile partition.
__device__ __forceinline__ int global_row(const FrameConst &fc, int local_row) {
    const int tile = local_row / fc.row_tile;
    return (tile * fc.num_parts + fc.part) * fc.row_tile + (local_row - tile * fc.row_tile);
}

// ---------------------------------------------------------------------------------------------
// Flyscene::sphereIntersection, src/flyscene.cpp:681-725.  pow(x, 2) promotes to double there.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool spheres_hit(const DeviceScene &sc, const Ray &r, float eps, float &tbest) {
    bool any = false;
    for (int i = 0; i < sc.num_spheres; ++i) {
        const float4 s = sc.spheres[i];
        const V3 oc = mk(s.x, s.y, s.z) - r.o;
        const float proj = dot(oc, r.d);
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

// ---------------------------------------------------------------------------------------------
// BVH walk.  ANY = false: closest hit with the reference's tie rule; ANY = true: first accepted
// face ends the walk (getLightIntensity only asks whether anything is hit, src/flyscene.cpp:884).
// `stack` points at this lane's column of the LDS stack (stride = workgroup size).
// ---------------------------------------------------------------------------------------------
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool walk(const DeviceScene &sc, const Ray &r, float eps, int *stack, int stride, float &tbest,
                                     int &best_tri, Work &wk) {
    // Slab test operands: t = fma(bound, inv, -origin * inv).  The inverse is kept finite (a zero
    // direction component becomes +-1e25) so that no infinity or NaN can enter the min/max chains.
    // The product origin*inv is rounded once, which puts an ABSOLUTE error of about one ulp of
    // |origin*inv| on every slab distance of that axis; `slack` bounds it (two ulps of the largest
    // product, clamped axes left out: their distances are exact zeros or astronomically large with a
    // reliable sign) and is added to the exit distance together with a relative widening.
    constexpr float kInvMax = 1e25f;
    float ix = 1.0f / r.d.x, iy = 1.0f / r.d.y, iz = 1.0f / r.d.z;
    const bool cx = !(fabsf(ix) <= kInvMax), cy = !(fabsf(iy) <= kInvMax), cz = !(fabsf(iz) <= kInvMax);
    if (cx) ix = copysignf(kInvMax, r.d.x);
    if (cy) iy = copysignf(kInvMax, r.d.y);
    if (cz) iz = copysignf(kInvMax, r.d.z);
    const float bx = -r.o.x * ix, by = -r.o.y * iy, bz = -r.o.z * iz;
    const float slack = 2.4e-7f * fmaxf(fmaxf(cx ? 0.0f : fabsf(bx), cy ? 0.0f : fabsf(by)), cz ? 0.0f : fabsf(bz));
    int best_rank = -1;
    bool found = false;
    int sp = 0;
    int node = 0;
    // far limit, widened so that equal and last-bit-different distances are still visited
    float tlimit = ANY ? FLT_MAX : tbest * 1.000001f;

    for (;;) {
        while (node >= 0 && node != kSentinel) {
            const float4 nx = sc.nodes[node * 4 + 0];
            const float4 ny = sc.nodes[node * 4 + 1];
            const float4 nz = sc.nodes[node * 4 + 2];
            const float4 nc = sc.nodes[node * 4 + 3];
            if (COUNT) wk.box += 2;
            // child 0
            const float a0 = __builtin_fmaf(nx.x, ix, bx), a1 = __builtin_fmaf(nx.y, ix, bx);
            const float b0 = __builtin_fmaf(ny.x, iy, by), b1 = __builtin_fmaf(ny.y, iy, by);
            const float c0 = __builtin_fmaf(nz.x, iz, bz), c1 = __builtin_fmaf(nz.y, iz, bz);
            const float n0 = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
            const float f0 = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), tlimit));
            // child 1
            const float d0 = __builtin_fmaf(nx.z, ix, bx), d1 = __builtin_fmaf(nx.w, ix, bx);
            const float e0 = __builtin_fmaf(ny.z, iy, by), e1 = __builtin_fmaf(ny.w, iy, by);
            const float g0 = __builtin_fmaf(nz.z, iz, bz), g1 = __builtin_fmaf(nz.w, iz, bz);
            const float n1 = fmaxf(fmaxf(fminf(d0, d1), fminf(e0, e1)), fmaxf(fminf(g0, g1), 0.0f));
            const float f1 = fminf(fminf(fmaxf(d0, d1), fmaxf(e0, e1)), fminf(fmaxf(g0, g1), tlimit));
            const bool h0 = n0 <= __builtin_fmaf(f0, 1.0000005f, slack);
            const bool h1 = n1 <= __builtin_fmaf(f1, 1.0000005f, slack);
            const int c0i = __float_as_int(nc.x), c1i = __float_as_int(nc.y);
            if (h0 && h1) {
                const bool first0 = n0 <= n1;
                stack[sp * stride] = first0 ? c1i : c0i;
                ++sp;
                node = first0 ? c0i : c1i;
            } else if (h0) {
                node = c0i;
            } else if (h1) {
                node = c1i;
            } else {
                if (sp == 0) {
                    node = kSentinel;
                } else {
                    --sp;
                    node = stack[sp * stride];
                }
            }
        }
        if (node == kSentinel) break;

        // leaf: exact plane + three-edge test of every face, src/flyscene.cpp:752-789
        const int ref = ~node;
        const int first = ref >> 4, count = ref & 15;
        for (int k = 0; k < count; ++k) {
            const int i = first + k;
            const float4 pn = sc.tris[i * 4 + 0];
            const float4 p0 = sc.tris[i * 4 + 1];
            const float4 p1 = sc.tris[i * 4 + 2];
            const float4 p2 = sc.tris[i * 4 + 3];
            if (COUNT) ++wk.plane;
            const V3 n = mk(pn.x, pn.y, pn.z);
            const float t = (pn.w - dot(r.o, n)) / dot(r.d, n);
            if (!(t >= eps && t <= tbest)) continue; // also rejects NaN (see file header)
            const int rank = __float_as_int(p1.w);
            if (!ANY && t == tbest && rank < best_rank) continue; // equal distance: the later face wins (:760)
            if (COUNT) ++wk.edge;
            const V3 v0 = mk(p0.x, p0.y, p0.z), v1 = mk(p1.x, p1.y, p1.z), v2 = mk(p2.x, p2.y, p2.z);
            const V3 hp = r.o + t * r.d;
            if (dot(cross(v1 - v0, hp - v0), n) < 0) continue;
            if (dot(cross(v2 - v1, hp - v1), n) < 0) continue;
            if (dot(cross(v0 - v2, hp - v2), n) < 0) continue;
            found = true;
            if (ANY) return true;
            tbest = t;
            best_tri = i;
            best_rank = rank;
            tlimit = t * 1.000001f;
        }
        if (sp == 0) break;
        --sp;
        node = stack[sp * stride];
    }
    return found;
}

__device__ __forceinline__ void flush_work(const Work &wk, WorkCounters *out) {
    unsigned a = wk.box, b = wk.plane, c = wk.edge;
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
        c += __shfl_xor(c, off);
    }
    if (lane_id() == 0) {
        atomicAdd(&out->box_tests, (unsigned long long)a);
        atomicAdd(&out->tri_plane, (unsigned long long)b);
        atomicAdd(&out->tri_edge, (unsigned long long)c);
    }
}

__device__ __forceinline__ float *plane(float *base, int capacity, int k) { return base + (size_t)k * (size_t)capacity; }

// ---------------------------------------------------------------------------------------------
// Closest-hit kernel.
//   PRIMARY: one workgroup = 16x16 pixels of the local image, one wave = an 8x8 tile; the grid is
//            remapped so that the workgroups sharing an XCD (blockIdx % 8) cover one horizontal band.
//   else:    one lane per entry of the current ray queue; the count is read from device memory, the
//            grid covers the worst case and surplus workgroups leave at once.
// Writes, per hit, (slot, record, hit point) at a position drawn with one atomic per wave.
// ---------------------------------------------------------------------------------------------
template <bool PRIMARY, bool COUNT>
__global__ __launch_bounds__(kTraceThreads) void trace_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int stage, int cur_queue,
                                                              float *rgb, int32_t *face_id, float *t_hit, int tiles_x,
                                                              int num_blocks) {
    __shared__ int s_stack[kStackEntries * kTraceThreads];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    int slot;
    bool valid;
    Ray ray;
    int pix_x = 0, pix_row = 0;
    if (PRIMARY) {
        // bijective XCD band remap: blocks b, b+8, b+16.. run on one XCD; give each XCD a contiguous range
        const int b = (int)blockIdx.x;
        const int q = num_blocks >> 3, rem = num_blocks & 7;
        const int xcd = b & 7, idx = b >> 3;
        const int blk = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
        const int bx = blk % tiles_x, by = blk / tiles_x;
        pix_x = bx * 16 + (wave & 1) * 8 + (lane & 7);
        pix_row = by * 16 + (wave >> 1) * 8 + (lane >> 3);
        valid = pix_x < fc.width && pix_row < fc.local_rows;
        slot = pix_row * fc.width + pix_x;
        ray.o = mk(fc.origin[0], fc.origin[1], fc.origin[2]);
        if (valid) {
            const int gy = global_row(fc, pix_row);
            ray.d = camera_direction(fc, (float)pix_x + fc.sub_x, (float)gy + fc.sub_y);
        } else {
            ray.d = mk(0.0f, 0.0f, -1.0f);
        }
    } else {
        const int count = ws.stage[stage - 1].next;
        if ((int)blockIdx.x * kTraceThreads >= count) return;
        slot = (int)blockIdx.x * kTraceThreads + tid;
        valid = slot < count;
        const int s = valid ? slot : 0;
        float *q = ws.queue[cur_queue];
        ray.o = mk(plane(q, ws.capacity, 0)[s], plane(q, ws.capacity, 1)[s], plane(q, ws.capacity, 2)[s]);
        ray.d = mk(plane(q, ws.capacity, 3)[s], plane(q, ws.capacity, 4)[s], plane(q, ws.capacity, 5)[s]);
    }

    float tbest = FLT_MAX;
    int best_tri = -2;
    Work wk = {0, 0, 0};
    if (valid) {
        // intersects(): spheres first, then triangles, sharing the running maximum (:673-676)
        if (sc.num_spheres > 0 && spheres_hit(sc, ray, fc.t_eps, tbest)) best_tri = -1;
        walk<false, COUNT>(sc, ray, fc.t_eps, s_stack + tid, kTraceThreads, tbest, best_tri, wk);
    }
    const bool hit = valid && best_tri != -2;

    if (PRIMARY && valid) {
        if (face_id) face_id[slot] = best_tri >= 0 ? __float_as_int(sc.tris[best_tri * 4 + 1].w) : best_tri;
        if (t_hit) t_hit[slot] = hit ? tbest : 0.0f;
        if (!hit) { // background of a primary ray, src/flyscene.cpp:653-658
            float *p = rgb + 3 * (size_t)slot;
            if (fc.accumulate) {
                p[0] += fc.bg[0], p[1] += fc.bg[1], p[2] += fc.bg[2];
            } else {
                p[0] = fc.bg[0], p[1] = fc.bg[1], p[2] = fc.bg[2];
            }
        }
    }
    if (!PRIMARY && valid && !hit) { // deeper miss adds throughput * background_deep (zero in the reference, :661-665)
        if (fc.bg_deep[0] != 0.0f || fc.bg_deep[1] != 0.0f || fc.bg_deep[2] != 0.0f) {
            float *q = ws.queue[cur_queue];
            const int pix = __float_as_int(plane(q, ws.capacity, 9)[slot]);
            float *p = rgb + 3 * (size_t)pix;
            p[0] += plane(q, ws.capacity, 6)[slot] * fc.bg_deep[0];
            p[1] += plane(q, ws.capacity, 7)[slot] * fc.bg_deep[1];
            p[2] += plane(q, ws.capacity, 8)[slot] * fc.bg_deep[2];
        }
    }

    // append hits: one atomic per wave, lanes keep their relative order
    const unsigned long long m = __ballot(hit);
    if (m != 0ull) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&ws.stage[stage].hits, __popcll(m));
        base = __shfl(base, 0);
        if (hit) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            const V3 hp = ray.o + tbest * ray.d; // hitPoint, :709,771
            ws.hit_slot[pos] = slot;
            ws.hit_tri[pos] = best_tri;
            plane(ws.hit_p, ws.capacity, 0)[pos] = hp.x;
            plane(ws.hit_p, ws.capacity, 1)[pos] = hp.y;
            plane(ws.hit_p, ws.capacity, 2)[pos] = hp.z;
        }
    }
    if (COUNT) flush_work(wk, ws.work);
}

// ---------------------------------------------------------------------------------------------
// Shading kernel: one workgroup (kShadeWaves waves) per 64 entries of the hit list.
// ---------------------------------------------------------------------------------------------
template <bool PRIMARY, bool COUNT>
__global__ __launch_bounds__(kShadeWaves * 64) void shade_kernel(DeviceScene sc, FrameConst fc, Workspace ws, int stage, int cur_queue,
                                                                 int spawn, float *rgb) {
    constexpr int kThreads = kShadeWaves * 64;
    __shared__ int s_stack[kStackEntries * kThreads];
    __shared__ unsigned long long s_mask[kMaxShadowRays];

    const int nhits = ws.stage[stage].hits;
    const int chunk0 = (int)blockIdx.x * 64;
    if (chunk0 >= nhits) return;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int idx = chunk0 + lane;
    const bool valid = idx < nhits;
    const int hidx = valid ? idx : chunk0;
    const V3 P = mk(plane(ws.hit_p, ws.capacity, 0)[hidx], plane(ws.hit_p, ws.capacity, 1)[hidx], plane(ws.hit_p, ws.capacity, 2)[hidx]);

    // --- getLightIntensity, src/flyscene.cpp:870-896: the sample POSITION is the ray direction (:884)
    const int nshadow = sc.num_lights * sc.spl;
    Work wk = {0, 0, 0};
    for (int j = wave; j < nshadow; j += kShadeWaves) {
        Ray sr;
        sr.o = P;
        sr.d = mk(sc.samples[3 * j], sc.samples[3 * j + 1], sc.samples[3 * j + 2]);
        bool occluded = false;
        if (valid) {
            float tb = FLT_MAX;
            int bt = -2;
            if (sc.num_spheres > 0 && spheres_hit(sc, sr, fc.t_eps, tb)) occluded = true;
            if (!occluded) occluded = walk<true, COUNT>(sc, sr, fc.t_eps, s_stack + tid, kThreads, tb, bt, wk);
        }
        const unsigned long long clear = __ballot(valid && !occluded);
        if (lane == 0) s_mask[j] = clear;
    }
    if (COUNT) flush_work(wk, ws.work + 1);
    __syncthreads();
    if (wave != 0 || !valid) return;

    float li = 0.0f;
    for (int l = 0; l < sc.num_lights; ++l) {
        float reaching = 0.0f;
        for (int k = 0; k < sc.spl; ++k) reaching += (float)((s_mask[l * sc.spl + k] >> lane) & 1ull);
        li += reaching / (float)sc.spl;
    }
    li = smin(li, 1.0f);
    const V3 LI = mk(li, li, li);

    // --- the ray that produced this hit
    const int slot = ws.hit_slot[idx];
    const int tri = ws.hit_tri[idx];
    V3 ro, rd, T;
    int pix;
    if (PRIMARY) {
        pix = slot;
        const int px = slot % fc.width, prow = slot / fc.width;
        ro = mk(fc.origin[0], fc.origin[1], fc.origin[2]);
        rd = camera_direction(fc, (float)px + fc.sub_x, (float)global_row(fc, prow) + fc.sub_y);
        T = mk(1.0f, 1.0f, 1.0f);
    } else {
        float *q = ws.queue[cur_queue];
        ro = mk(plane(q, ws.capacity, 0)[slot], plane(q, ws.capacity, 1)[slot], plane(q, ws.capacity, 2)[slot]);
        rd = mk(plane(q, ws.capacity, 3)[slot], plane(q, ws.capacity, 4)[slot], plane(q, ws.capacity, 5)[slot]);
        T = mk(plane(q, ws.capacity, 6)[slot], plane(q, ws.capacity, 7)[slot], plane(q, ws.capacity, 8)[slot]);
        pix = __float_as_int(plane(q, ws.capacity, 9)[slot]);
    }

    // --- shadeOffFace, src/flyscene.cpp:461-579
    V3 local;
    bool reflect = false;
    V3 ks = mk(0, 0, 0), fn = mk(0, 0, 0);
    if (tri < 0) {
        local = mk(0.7f, 0.9f, 0.0f); // sphere, :463-465
    } else {
        const int face = __float_as_int(sc.tris[tri * 4 + 1].w);
        const float4 s0 = sc.shade[face * 6 + 0], s1 = sc.shade[face * 6 + 1], s2 = sc.shade[face * 6 + 2];
        const float4 s3 = sc.shade[face * 6 + 3], s4 = sc.shade[face * 6 + 4], s5 = sc.shade[face * 6 + 5];
        const V3 v0 = mk(s0.x, s0.y, s0.z), v1 = mk(s0.w, s1.x, s1.y), v2 = mk(s1.z, s1.w, s2.x);
        const V3 n0 = mk(s2.y, s2.z, s2.w), n1 = mk(s3.x, s3.y, s3.z), n2 = mk(s3.w, s4.x, s4.y);
        fn = mk(s4.z, s4.w, s5.x);
        const int mat = __float_as_int(s5.y);
        if (mat < 0 || mat >= sc.num_materials) {
            local = mk(0.5f, 0.5f, 0.5f); // :469-472
        } else {
            const float *mp = sc.materials + 10 * mat;
            const V3 ka = mk(mp[0], mp[1], mp[2]), kd = mk(mp[3], mp[4], mp[5]);
            ks = mk(mp[6], mp[7], mp[8]);
            const float shininess = mp[9];
            // area-weighted normal, :494-500
            const float a12 = norm(cross(v1 - P, v2 - P)) * 0.5f;
            const float a02 = norm(cross(v0 - P, v2 - P)) * 0.5f;
            const float a01 = norm(cross(v0 - P, v1 - P)) * 0.5f;
            const V3 N = normalized((a12 * n0 + a02 * n1) + a01 * n2);
            local = mk(0.0f, 0.0f, 0.0f);
            for (int l = 0; l < sc.num_lights; ++l) {
                const V3 lp = mk(sc.light_pos[3 * l], sc.light_pos[3 * l + 1], sc.light_pos[3 * l + 2]);
                const V3 L = normalized(lp - P);                                   // :507
                const V3 ambient = cmul(LI, ka);                                   // :510
                const float cos1 = fmaxf(0.0f, dot(L, N));                         // :513
                const V3 diffuse = cos1 * cmul(LI, kd);                            // :514
                const V3 E = normalized(ro - P);                                   // :545
                const V3 R = normalized(-L + (2.f * dot(L, N)) * N);               // :546
                const float cos2 = fmaxf(0.0f, dot(R, E));                         // :548
                const V3 specular = powf(cos2, shininess) * cmul(LI, ks);          // :549
                const V3 sum = (diffuse + specular) + ambient;                     // :554
                local = local + mk(smin(smax(sum.x, 0.0f), 1.0f), smin(smax(sum.y, 0.0f), 1.0f), smin(smax(sum.z, 0.0f), 1.0f));
            }
            reflect = ks.x > fc.refl_thresh || ks.y > fc.refl_thresh || ks.z > fc.refl_thresh; // :631-632
        }
    }

    // pixel += throughput * local.  Each pixel owns at most one live path, so no atomics are needed.
    float *p = rgb + 3 * (size_t)pix;
    const V3 add = cmul(T, local);
    if (PRIMARY && !fc.accumulate) {
        p[0] = local.x, p[1] = local.y, p[2] = local.z;
    } else {
        p[0] += add.x, p[1] += add.y, p[2] += add.z;
    }

    // --- reflection set-up of traceRay, src/flyscene.cpp:617-646
    const bool go = spawn && reflect;
    const unsigned long long m = __ballot(go);
    if (m != 0ull) {
        int base = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) base = atomicAdd(&ws.stage[stage].next, __popcll(m));
        base = __shfl(base, leader);
        if (go) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            const V3 refl = normalized(rd - (2 * dot(rd, fn)) * fn); // :785 then :635
            const V3 no = P + fc.bounce_off * refl;                  // :636
            const V3 nT = cmul(cmul(T, ks), LI);                     // :637
            float *q = ws.queue[cur_queue ^ 1];
            plane(q, ws.capacity, 0)[pos] = no.x, plane(q, ws.capacity, 1)[pos] = no.y, plane(q, ws.capacity, 2)[pos] = no.z;
            plane(q, ws.capacity, 3)[pos] = refl.x, plane(q, ws.capacity, 4)[pos] = refl.y, plane(q, ws.capacity, 5)[pos] = refl.z;
            plane(q, ws.capacity, 6)[pos] = nT.x, plane(q, ws.capacity, 7)[pos] = nT.y, plane(q, ws.capacity, 8)[pos] = nT.z;
            plane(q, ws.capacity, 9)[pos] = __int_as_float(pix);
        }
    }
}

// pixel /= SSAALEVEL^2, src/flyscene.cpp:830
__global__ void scale_kernel(float *rgb, size_t n, float divisor) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rgb[i] = rgb[i] / divisor;
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
void launch_trace(bool primary, bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int cur_queue,
                  float *rgb, int32_t *face_id, float *t_hit, hipStream_t stream) {
    if (primary) {
        const int tiles_x = (fc.width + 15) / 16, tiles_y = (fc.local_rows + 15) / 16;
        const int blocks = tiles_x * tiles_y;
        if (blocks <= 0) return;
        if (count)
            trace_kernel<true, true><<<blocks, kTraceThreads, 0, stream>>>(sc, fc, ws, stage, cur_queue, rgb, face_id, t_hit, tiles_x, blocks);
        else
            trace_kernel<true, false><<<blocks, kTraceThreads, 0, stream>>>(sc, fc, ws, stage, cur_queue, rgb, face_id, t_hit, tiles_x, blocks);
    } else {
        const int blocks = (ws.capacity + kTraceThreads - 1) / kTraceThreads;
        if (blocks <= 0) return;
        if (count)
            trace_kernel<false, true><<<blocks, kTraceThreads, 0, stream>>>(sc, fc, ws, stage, cur_queue, rgb, face_id, t_hit, 0, blocks);
        else
            trace_kernel<false, false><<<blocks, kTraceThreads, 0, stream>>>(sc, fc, ws, stage, cur_queue, rgb, face_id, t_hit, 0, blocks);
    }
}

void launch_shade(bool primary, bool count, const DeviceScene &sc, const FrameConst &fc, const Workspace &ws, int stage, int cur_queue,
                  int spawn, float *rgb, hipStream_t stream) {
    const int blocks = (ws.capacity + 63) / 64;
    if (blocks <= 0) return;
    constexpr int threads = kShadeWaves * 64;
    if (primary) {
        if (count)
            shade_kernel<true, true><<<blocks, threads, 0, stream>>>(sc, fc, ws, stage, cur_queue, spawn, rgb);
        else
            shade_kernel<true, false><<<blocks, threads, 0, stream>>>(sc, fc, ws, stage, cur_queue, spawn, rgb);
    } else {
        if (count)
            shade_kernel<false, true><<<blocks, threads, 0, stream>>>(sc, fc, ws, stage, cur_queue, spawn, rgb);
        else
            shade_kernel<false, false><<<blocks, threads, 0, stream>>>(sc, fc, ws, stage, cur_queue, spawn, rgb);
    }
}

void launch_scale(float *rgb, size_t n, float divisor, hipStream_t stream) {
    if (n == 0) return;
    scale_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(rgb, n, divisor);
}

void launch_assemble(const void *src, void *dst, int width, int height, int channels, int row_tile, int num_parts, hipStream_t stream) {
    const size_t n = (size_t)width * height * channels;
    if (n == 0) return;
    assemble_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>((const uint32_t *)src, (uint32_t *)dst, width, height, channels, row_tile,
                                                                      num_parts);
}

} // namespace rtamd
