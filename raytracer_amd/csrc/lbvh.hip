#include "lbvh.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "device_types.h"

namespace rtamd {

namespace {

struct Box3 {
    float lo[3], hi[3];
};

__device__ __forceinline__ unsigned long long spread21(unsigned long long v) { // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__device__ __forceinline__ Box3 face_box(const TriRecord &t) {
    Box3 b;
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = fminf(fminf(t.v0[k], t.v1[k]), t.v2[k]);
        b.hi[k] = fmaxf(fmaxf(t.v0[k], t.v1[k]), t.v2[k]);
    }
    return b;
}

__global__ void morton_kernel(const TriRecord *faces, int n, Box3 scene, unsigned long long *keys, int *values) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const Box3 b = face_box(faces[i]);
    unsigned long long q[3];
    for (int k = 0; k < 3; ++k) {
        const float ext = scene.hi[k] - scene.lo[k];
        const float c = 0.5f * (b.lo[k] + b.hi[k]);
        float u = ext > 0.0f ? (c - scene.lo[k]) / ext : 0.0f;
        u = fminf(fmaxf(u * 2097152.0f, 0.0f), 2097151.0f); // 21 bits per axis: a face in a large scene still gets a cell of its own
        q[k] = (unsigned long long)u;
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    values[i] = i;
}

__global__ void gather_kernel(const TriRecord *faces, const int *perm, int n, TriRecord *sorted, Box3 *leaf_box) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const float4 *src = (const float4 *)(faces + perm[i]);
    float4 *dst = (float4 *)(sorted + i);
    dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
    leaf_box[i] = face_box(sorted[i]);
}

// length of the common prefix of keys i and j (ties broken by the position), -1 outside the array
__device__ __forceinline__ int common_prefix(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((unsigned)i ^ (unsigned)j);
}

// Internal node i of the binary radix tree over the sorted keys: children as references (>= 0 internal node,
// < 0 leaf ~position), the sorted positions [first, last] it spans, parents of both.
__global__ void radix_tree_kernel(const unsigned long long *keys, int n, int *child, int2 *range, int *parent_of_node, int *parent_of_leaf) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = common_prefix(keys, n, i, i - d);
    int lmax = 2;
    while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (common_prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = common_prefix(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (common_prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int split = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int left = split == first ? ~split : split;
    const int right = split + 1 == last ? ~(split + 1) : split + 1;
    child[2 * i] = left, child[2 * i + 1] = right;
    range[i] = make_int2(first, last);
    if (left >= 0) parent_of_node[left] = i; else parent_of_leaf[~left] = i;
    if (right >= 0) parent_of_node[right] = i; else parent_of_leaf[~right] = i;
    if (i == 0) parent_of_node[0] = -1;
}

__device__ __forceinline__ Box3 merged(const Box3 &a, const Box3 &b) {
    Box3 m;
    for (int k = 0; k < 3; ++k) m.lo[k] = fminf(a.lo[k], b.lo[k]), m.hi[k] = fmaxf(a.hi[k], b.hi[k]);
    return m;
}

// Boxes bottom-up: the second thread to arrive at a node fits it and goes on; the first one stops.
__global__ void fit_kernel(int n, const int *child, const int *parent_of_node, const int *parent_of_leaf, const Box3 *leaf_box, Box3 *node_box,
                           int *arrivals) {
    const int leaf = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (leaf >= n) return;
    int node = parent_of_leaf[leaf];
    while (node >= 0) {
        __threadfence(); // the box written below is visible before the arrival is
        if (atomicAdd(&arrivals[node], 1) == 0) return;
        __threadfence(); // and the sibling's box is read after its arrival was seen
        const int l = child[2 * node], r = child[2 * node + 1];
        const Box3 bl = l >= 0 ? node_box[l] : leaf_box[~l];
        const Box3 br = r >= 0 ? node_box[r] : leaf_box[~r];
        node_box[node] = merged(bl, br);
        node = parent_of_node[node];
    }
}

// depth of every leaf (edges to the root), maximum over all
__global__ void depth_kernel(int n, const int *parent_of_node, const int *parent_of_leaf, int *max_depth) {
    const int leaf = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (leaf >= n) return;
    int depth = 0;
    for (int node = parent_of_leaf[leaf]; node >= 0; node = parent_of_node[node]) ++depth;
    atomicMax(max_depth, depth);
}

// Four-way node of binary node i: its children, each replaced by ITS two children when it is an inner node.
// References to inner nodes keep the binary numbering (only every other level is ever referenced).  A
// subtree spanning at most max_leaf sorted positions becomes one leaf over that run of records.
__global__ void collapse_kernel(int n, int max_leaf, const int *child, const int2 *range, const Box3 *leaf_box, const Box3 *node_box, Bvh4Node *out) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    auto small = [&](int r) { return r >= 0 && range[r].y - range[r].x < max_leaf; };
    // Start from the two children and, twice, replace the expandable child of largest surface area by ITS two
    // children: the large boxes are the ones a ray is likely to enter anyway, so opening them here saves a step.
    // (Any binary node may so become a child of a four-way node; every binary node has a four-way node written.)
    auto expandable = [&](int r) { return r >= 0 && !small(r); };
    auto area = [&](int r) {
        const Box3 &b = node_box[r];
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    int refs[4];
    int k = 2;
    refs[0] = child[2 * i], refs[1] = child[2 * i + 1];
    while (k < 4) {
        int pick = -1;
        float best = -1.0f;
        for (int s = 0; s < k; ++s)
            if (expandable(refs[s])) {
                const float a = area(refs[s]);
                if (a > best) best = a, pick = s;
            }
        if (pick < 0) break;
        const int c = refs[pick];
        refs[pick] = child[2 * c];
        refs[k++] = child[2 * c + 1];
    }
    Bvh4Node nd;
    for (int s = 0; s < 4; ++s) {
        if (s < k) {
            const int r = refs[s];
            const Box3 b = r >= 0 ? node_box[r] : leaf_box[~r];
            nd.lox[s] = b.lo[0], nd.hix[s] = b.hi[0], nd.loy[s] = b.lo[1], nd.hiy[s] = b.hi[1], nd.loz[s] = b.lo[2], nd.hiz[s] = b.hi[2];
            if (r < 0)
                nd.child[s] = ~(((~r) << 4) | 1);
            else if (small(r))
                nd.child[s] = ~((range[r].x << 4) | (range[r].y - range[r].x + 1));
            else
                nd.child[s] = r;
        } else {
            nd.lox[s] = nd.loy[s] = nd.loz[s] = FLT_MAX;
            nd.hix[s] = nd.hiy[s] = nd.hiz[s] = -FLT_MAX;
            nd.child[s] = ~0;
        }
    }
    nd.num_children = k;
    nd.pad[0] = nd.pad[1] = nd.pad[2] = 0;
    out[i] = nd;
}

struct DeviceBuffers {
    std::vector<void *> all;
    ~DeviceBuffers() {
        for (void *p : all)
            if (p) (void)hipFree(p);
    }
    template <class T>
    bool get(T **p, size_t count) {
        void *raw = nullptr;
        if (hipMalloc(&raw, count * sizeof(T) + 64) != hipSuccess) return false;
        all.push_back(raw);
        *p = (T *)raw;
        return true;
    }
};

} // namespace

bool build_lbvh_device(const std::vector<TriRecord> &faces, int max_leaf_faces, FlatBvh &out, std::string &why) {
    const int n = (int)faces.size();
    if (n < 2) {
        why = "fewer than two faces";
        return false;
    }
    if (max_leaf_faces <= 0) max_leaf_faces = 4;
    max_leaf_faces = std::min(max_leaf_faces, kMaxLeafFaces);
    Box3 scene;
    for (int k = 0; k < 3; ++k) scene.lo[k] = FLT_MAX, scene.hi[k] = -FLT_MAX;
    for (const TriRecord &t : faces)
        for (int k = 0; k < 3; ++k) {
            scene.lo[k] = std::min(scene.lo[k], std::min(t.v0[k], std::min(t.v1[k], t.v2[k])));
            scene.hi[k] = std::max(scene.hi[k], std::max(t.v0[k], std::max(t.v1[k], t.v2[k])));
        }
#define LBVH_TRY(expr)                                                     \
    do {                                                                   \
        const hipError_t e_ = (expr);                                      \
        if (e_ != hipSuccess) {                                            \
            why = std::string(#expr " failed: ") + hipGetErrorString(e_);  \
            return false;                                                  \
        }                                                                  \
    } while (0)

    const bool verbose = std::getenv("RTAMD_VERBOSE_BUILD") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    DeviceBuffers buf;
    TriRecord *d_faces = nullptr, *d_sorted = nullptr;
    unsigned long long *d_keys = nullptr, *d_keys_sorted = nullptr;
    int *d_vals = nullptr, *d_perm = nullptr, *d_child = nullptr, *d_parent_node = nullptr, *d_parent_leaf = nullptr, *d_arrivals = nullptr,
        *d_depth = nullptr;
    Box3 *d_leaf_box = nullptr, *d_node_box = nullptr;
    int2 *d_range = nullptr;
    Bvh4Node *d_nodes4 = nullptr;
    const size_t N = (size_t)n;
    if (!buf.get(&d_faces, N) || !buf.get(&d_sorted, N) || !buf.get(&d_keys, N) || !buf.get(&d_keys_sorted, N) || !buf.get(&d_vals, N) ||
        !buf.get(&d_perm, N) || !buf.get(&d_child, 2 * N) || !buf.get(&d_range, N) || !buf.get(&d_parent_node, N) || !buf.get(&d_parent_leaf, N) ||
        !buf.get(&d_arrivals, N) || !buf.get(&d_depth, 1) || !buf.get(&d_leaf_box, N) || !buf.get(&d_node_box, N) || !buf.get(&d_nodes4, N)) {
        why = "out of device memory";
        return false;
    }
    const double t1 = now();
    LBVH_TRY(hipMemcpy(d_faces, faces.data(), N * sizeof(TriRecord), hipMemcpyHostToDevice));
    const double t2 = now();
    const unsigned blocks = (unsigned)((n + 255) / 256);
    morton_kernel<<<blocks, 256>>>(d_faces, n, scene, d_keys, d_vals);
    size_t temp_bytes = 0;
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_vals, d_perm, n, 0, 63));
    char *d_temp = nullptr;
    if (!buf.get(&d_temp, temp_bytes)) {
        why = "out of device memory";
        return false;
    }
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_keys_sorted, d_vals, d_perm, n, 0, 63));
    gather_kernel<<<blocks, 256>>>(d_faces, d_perm, n, d_sorted, d_leaf_box);
    radix_tree_kernel<<<blocks, 256>>>(d_keys_sorted, n, d_child, d_range, d_parent_node, d_parent_leaf);
    LBVH_TRY(hipMemset(d_arrivals, 0, N * sizeof(int)));
    LBVH_TRY(hipMemset(d_depth, 0, sizeof(int)));
    fit_kernel<<<blocks, 256>>>(n, d_child, d_parent_node, d_parent_leaf, d_leaf_box, d_node_box, d_arrivals);
    depth_kernel<<<blocks, 256>>>(n, d_parent_node, d_parent_leaf, d_depth);
    collapse_kernel<<<blocks, 256>>>(n, max_leaf_faces, d_child, d_range, d_leaf_box, d_node_box, d_nodes4);
    LBVH_TRY(hipGetLastError());
    int depth2 = 0;
    LBVH_TRY(hipMemcpy(&depth2, d_depth, sizeof(int), hipMemcpyDeviceToHost));
    const double t3 = now();
    out.nodes.clear();
    out.nodes4.resize(N - 1);
    out.tris.resize(N);
    LBVH_TRY(hipMemcpy(out.nodes4.data(), d_nodes4, (N - 1) * sizeof(Bvh4Node), hipMemcpyDeviceToHost));
    LBVH_TRY(hipMemcpy(out.tris.data(), d_sorted, N * sizeof(TriRecord), hipMemcpyDeviceToHost));
    // Most entries a depth-first walk can hold: at a node with k children one is entered and k - 1 wait, whatever
    // the order; the deepest such chain decides.  (Long Morton codes make deep but thin trees: three entries per
    // level, the bound without looking, would turn many of them down.)
    struct Need {
        const std::vector<Bvh4Node> &nodes;
        int deepest = 0;
        int of(int32_t node, int level) {
            if (level > 128) return 1 << 20; // not a tree this walk can take
            deepest = std::max(deepest, level + 1);
            const Bvh4Node &nd = nodes[(size_t)node];
            int below = 0;
            for (int c = 0; c < nd.num_children; ++c)
                if (nd.child[c] >= 0) below = std::max(below, of(nd.child[c], level + 1));
            return std::max(nd.num_children - 1, 0) + below;
        }
    };
    Need walk{out.nodes4};
    const int need = walk.of(0, 0);
    if (need > kStackEntries) {
        why = "radix tree of depth " + std::to_string(depth2) + " needs " + std::to_string(need) + " stack entries";
        return false;
    }
    if (verbose)
        std::fprintf(stderr, "rtamd: device builder, %d faces: allocate %.2f ms, upload %.2f ms, kernels %.2f ms, download %.2f ms\n", n, t1 - t0,
                     t2 - t1, t3 - t2, now() - t3);
    out.num_leaves = n;
    out.depth = depth2;
    out.depth4 = walk.deepest;
    out.stack_need = need;
    out.num_device_faces = n;
    return true;
#undef LBVH_TRY
}

} // namespace rtamd
