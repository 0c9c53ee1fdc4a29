// Host-side BVH construction for the gfx950 render pass.
//
// The reference keeps a tree of heavyweight boundingBox objects (src/boundingBox.h:25-316) and
// visits every leaf a ray's boxes overlap.  Here the hierarchy is a flat array of 128-byte nodes,
// each holding the boxes of up to FOUR children so that one fetch decides the next step and the chain
// of dependent fetches per ray is half as long as in a binary tree, plus a leaf-contiguous array of
// 64-byte triangle records.  The tree is a binned-SAH binary tree collapsed to four-way: because
// the reference's walk is exhaustive, any conservative hierarchy yields the same closest hit; the
// only order-dependent rule of the reference (equal distances: the face met later wins,
// src/flyscene.cpp:760,782-789) is carried as an explicit per-face rank, computed by replaying
// boundingBox::splitBox (src/boundingBox.h:69-194).
#pragma once
#include <cstdint>
#include <vector>

namespace rtamd {

// 64 B.  Per axis: (child0.min, child0.max, child1.min, child1.max).  child < 0 encodes a leaf:
// ~child = (first_record << 4) | count.
struct alignas(64) BvhNode {
    float x[4];
    float y[4];
    float z[4];
    int32_t child[2];
    int32_t pad[2];
};

// 128 B, the device node.  Children's boxes as structure of arrays; child as in BvhNode.  Unused slots
// hold an inverted box (lo = +FLT_MAX, hi = -FLT_MAX), which the sign-ordered slab test never enters,
// and the reference of a leaf without records.
struct alignas(128) Bvh4Node {
    float lox[4], hix[4];
    float loy[4], hiy[4];
    float loz[4], hiz[4];
    int32_t child[4];
    int32_t num_children;
    int32_t pad[3];
};

// 64 B.  What triangleIntersection reads per face (src/flyscene.cpp:752-789), gathered once.
struct alignas(64) TriRecord {
    float n[3];   // faceNormals[i]
    float d;      // faceOriginDistance[i]
    float v0[3];
    int32_t face; // index in the caller's face arrays
    float v1[3];
    int32_t rank; // position in the reference's face visiting order
    float v2[3];
    int32_t pad;
};

// 128 B, one per TriRecord and in the same order.  What shadeOffFace reads (src/flyscene.cpp:469-502).
struct alignas(128) ShadeRecord {
    float v0[3], v1[3], v2[3];
    float n0[3], n1[3], n2[3];
    float fn[3];
    int32_t material;
    float tc[6]; // Face::texcoord: (u,v) of the three corners
    int32_t pad[4];
};
static_assert(sizeof(ShadeRecord) == 128, "kernels index ShadeRecord as 8 float4");

constexpr int kMaxLeafFaces = 15;
constexpr int kMaxTreeDepth = 32; // of the binary tree

struct FlatBvh {
    std::vector<BvhNode> nodes;   // binary tree, nodes[0] is the root (host only)
    std::vector<Bvh4Node> nodes4; // what the kernels walk, nodes4[0] is the root
    std::vector<TriRecord> tris;
    int num_leaves = 0;
    int depth = 0;      // binary tree
    int depth4 = 0;     // collapsed tree
    int stack_need = 0; // most entries a depth-first walk of nodes4 can hold at once
    int num_device_faces = 0;
};

struct MeshView {
    const float *vertices;
    int32_t num_vertices;
    const int32_t *face_vertex_ids;
    const float *face_normals;
    const float *face_origin_distance;
    int32_t num_faces;
};

// rank[f] = position of face f in the sequence in which the reference's triangleIntersection would
// meet the faces if every leaf were hit: leaves in (lower, upper) preorder of the tree splitBox
// builds, faces in the order splitBox left them in each leaf.
void reference_face_rank(const MeshView &mesh, std::vector<int32_t> &rank);

// Binned-SAH build over the faces whose stored normal is finite and non-zero.
void build_flat_bvh(const MeshView &mesh, const std::vector<int32_t> &rank, int max_leaf_faces, int sah_bins, FlatBvh &out);

// The records of the faces build_flat_bvh would keep, in face order (input of the on-GPU builder).
void gather_face_records(const MeshView &mesh, const std::vector<int32_t> &rank, std::vector<TriRecord> &out);

// nodes -> nodes4 with at most `max_children` (2..4) children per node: repeatedly replaces the inner
// child of largest surface area by its two children.  Fills depth4 and stack_need.
void collapse_bvh(FlatBvh &bvh, int max_children);

} // namespace rtamd
