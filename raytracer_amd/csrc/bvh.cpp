#include "bvh.hpp"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>
#include <thread>

namespace rtamd {

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        lo[0] = lo[1] = lo[2] = FLT_MAX;
        hi[0] = hi[1] = hi[2] = -FLT_MAX;
    }
    void grow(const float *p) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box &b) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

inline const float *vtx(const MeshView &m, int32_t id) { return m.vertices + 3 * (size_t)id; }

} // namespace

// ------------------------------------------------------------------------------------------------
// Replay of boundingBox::splitBox (src/boundingBox.h:69-194) on index ranges.  Only the ORDER in
// which faces end up in the leaves is kept.  All comparisons and the float accumulations follow
// the reference so that the straddler alternation (:134-137) deals faces to the same side.
// ------------------------------------------------------------------------------------------------
namespace {

struct RankBuilder {
    const MeshView &m;
    std::vector<int32_t> &rank;

    // Every face lands in exactly one leaf, so the faces below a node take the ranks base .. base + n - 1
    // whatever happens further down: the two halves of a large node can be replayed by different threads.
    void split(const std::vector<int32_t> &faces, const float *bmin, const float *bmax, int32_t base, int forks_left) {
        const size_t n = faces.size();
        if (n < 10) { // baseCase, boundingBox.h:34,73
            for (int32_t f : faces) rank[f] = base++;
            return;
        }
        const float side[3] = {bmax[0] - bmin[0], bmax[1] - bmin[1], bmax[2] - bmin[2]};
        const int axis = (int)(std::max_element(side, side + 3) - side);
        float sum = 0.0f, weight = 0.0f;
        for (int32_t f : faces) {
            const int32_t *id = m.face_vertex_ids + 3 * (size_t)f;
            sum += vtx(m, id[0])[axis] + vtx(m, id[1])[axis] + vtx(m, id[2])[axis];
            weight += 3.0f;
        }
        const float avg = sum / weight;

        std::vector<int32_t> lower, upper;
        lower.reserve(n);
        upper.reserve(n);
        // child boxes start inverted from the parent's corners (:115-119)
        float upMin[3] = {bmax[0], bmax[1], bmax[2]}, upMax[3] = {bmin[0], bmin[1], bmin[2]};
        float loMin[3] = {bmax[0], bmax[1], bmax[2]}, loMax[3] = {bmin[0], bmin[1], bmin[2]};
        unsigned straddlers = 0;
        for (int32_t f : faces) {
            const int32_t *id = m.face_vertex_ids + 3 * (size_t)f;
            const float *p0 = vtx(m, id[0]), *p1 = vtx(m, id[1]), *p2 = vtx(m, id[2]);
            const bool in_upper = p0[axis] > avg && p1[axis] > avg && p2[axis] > avg;
            const bool in_lower = p0[axis] < avg && p1[axis] < avg && p2[axis] < avg;
            const bool to_upper = in_upper || (!in_lower && straddlers % 2 == 0);
            float *mn = to_upper ? upMin : loMin, *mx = to_upper ? upMax : loMax;
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(p0[a], std::min(p1[a], std::min(p2[a], mn[a])));
                mx[a] = std::max(p0[a], std::max(p1[a], std::max(p2[a], mx[a])));
            }
            (to_upper ? upper : lower).push_back(f);
            if (to_upper ? !in_upper : !in_lower) ++straddlers;
        }
        if ((double)lower.size() < 0.8 * (double)n && (double)upper.size() < 0.8 * (double)n) {
            const int32_t upper_base = base + (int32_t)lower.size();
            if (forks_left > 0 && n >= 32768) {
                std::thread other([&] { split(lower, loMin, loMax, base, forks_left - 1); });
                split(upper, upMin, upMax, upper_base, forks_left - 1);
                other.join();
            } else {
                split(lower, loMin, loMax, base, 0);
                split(upper, upMin, upMax, upper_base, 0);
            }
        } else {
            for (int32_t f : faces) rank[f] = base++;
        }
    }
};

} // namespace

void reference_face_rank(const MeshView &mesh, std::vector<int32_t> &rank) {
    rank.assign((size_t)mesh.num_faces, 0);
    if (mesh.num_faces == 0) return;
    // Flyscene::boundingVectors, src/flyscene.cpp:841-868: over every entry of the vertex array
    float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
    if (mesh.num_vertices > 0) {
        for (int a = 0; a < 3; ++a) bmin[a] = bmax[a] = mesh.vertices[a];
        for (int32_t i = 0; i < mesh.num_vertices; ++i)
            for (int a = 0; a < 3; ++a) {
                bmin[a] = std::min(bmin[a], vtx(mesh, i)[a]);
                bmax[a] = std::max(bmax[a], vtx(mesh, i)[a]);
            }
    }
    std::vector<int32_t> all((size_t)mesh.num_faces);
    std::iota(all.begin(), all.end(), 0);
    RankBuilder rb{mesh, rank};
    rb.split(all, bmin, bmax, 0, 4); // up to 16 threads at the fourth level
}

// ------------------------------------------------------------------------------------------------
// Binned SAH build.
// ------------------------------------------------------------------------------------------------
namespace {

struct Prim {
    Box box;
    float c[3];
    int32_t face;
};

struct SahBuilder {
    const MeshView &m;
    const std::vector<int32_t> &rank;
    std::vector<Prim> prims;
    FlatBvh &out;
    int max_leaf;
    int bins;

    static constexpr float kTraversalCost = 1.0f;
    static constexpr float kTriCost = 2.0f; // the exact plane + three-edge test costs about two node visits

    std::atomic<int> leaves{0}, deepest{0};

    // Every prim ends in exactly one leaf and leaves keep the order of `prims`: record i belongs at position i.
    int32_t make_leaf(int lo, int hi, Box &box) {
        box.reset();
        const int32_t first = lo;
        for (int i = lo; i < hi; ++i) {
            const Prim &p = prims[i];
            box.grow(p.box);
            TriRecord t;
            std::memset(&t, 0, sizeof t);
            const int32_t f = p.face;
            const int32_t *id = m.face_vertex_ids + 3 * (size_t)f;
            std::memcpy(t.n, m.face_normals + 3 * (size_t)f, 12);
            t.d = m.face_origin_distance[f];
            std::memcpy(t.v0, vtx(m, id[0]), 12);
            std::memcpy(t.v1, vtx(m, id[1]), 12);
            std::memcpy(t.v2, vtx(m, id[2]), 12);
            t.face = f;
            t.rank = rank[f];
            out.tris[(size_t)i] = t;
        }
        leaves.fetch_add(1, std::memory_order_relaxed);
        return ~((first << 4) | (hi - lo));
    }

    // Returns the child reference (inner index into `nodes`, or encoded leaf) and the bounds of the subtree.
    // Large nodes hand their second half to another thread, which builds into a vector of its own; it is
    // appended behind the first half's nodes, so the node order is the one a serial build gives.
    int32_t build(int lo, int hi, int depth, Box &box, std::vector<BvhNode> &nodes, int forks_left) {
        const int n = hi - lo;
        for (int seen = deepest.load(std::memory_order_relaxed); seen < depth && !deepest.compare_exchange_weak(seen, depth);) {
        }
        // depth budget: how many levels a balanced split still needs
        int need = 0;
        while (((size_t)max_leaf << need) < (size_t)n) ++need;
        const int budget = kMaxTreeDepth - depth;
        if (n <= max_leaf && (n <= 1 || budget <= 0)) return make_leaf(lo, hi, box);

        Box cb, nb;
        cb.reset();
        nb.reset();
        for (int i = lo; i < hi; ++i) {
            cb.grow(prims[i].c);
            nb.grow(prims[i].box);
        }
        int best_axis = -1, best_bin = -1;
        float best_cost = FLT_MAX;
        const bool force_median = need >= budget; // no slack left: stay balanced
        if (!force_median) {
            std::vector<Box> bbox((size_t)bins), right((size_t)bins);
            std::vector<int> bcount((size_t)bins);
            for (int a = 0; a < 3; ++a) {
                const float ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0)) continue;
                for (int b = 0; b < bins; ++b) bbox[b].reset(), bcount[b] = 0;
                const float scale = (float)bins / ext;
                for (int i = lo; i < hi; ++i) {
                    int b = (int)((prims[i].c[a] - cb.lo[a]) * scale);
                    b = std::min(std::max(b, 0), bins - 1);
                    bbox[b].grow(prims[i].box);
                    ++bcount[b];
                }
                Box acc;
                acc.reset();
                for (int b = bins - 1; b > 0; --b) {
                    acc.grow(bbox[b]);
                    right[b] = acc;
                }
                acc.reset();
                int nl = 0;
                for (int b = 0; b < bins - 1; ++b) {
                    acc.grow(bbox[b]);
                    nl += bcount[b];
                    const int nr = n - nl;
                    if (nl == 0 || nr == 0) continue;
                    const float cost = acc.half_area() * (float)nl + right[b + 1].half_area() * (float)nr;
                    if (cost < best_cost) best_cost = cost, best_axis = a, best_bin = b;
                }
            }
            if (n <= max_leaf) {
                if (best_axis < 0) return make_leaf(lo, hi, box);
                const float leaf_cost = kTriCost * (float)n * nb.half_area();
                const float split_cost = kTraversalCost * nb.half_area() + kTriCost * best_cost;
                if (leaf_cost <= split_cost) return make_leaf(lo, hi, box);
            }
        }
        int mid;
        if (best_axis >= 0) {
            const int a = best_axis;
            const float scale = (float)bins / (cb.hi[a] - cb.lo[a]);
            const float lo_a = cb.lo[a];
            const int bb = best_bin;
            const int nb_ = bins;
            auto it = std::partition(prims.begin() + lo, prims.begin() + hi, [=](const Prim &p) {
                int b = (int)((p.c[a] - lo_a) * scale);
                b = std::min(std::max(b, 0), nb_ - 1);
                return b <= bb;
            });
            mid = (int)(it - prims.begin());
        } else {
            // median split on the widest centroid axis (also the fallback when all centroids coincide)
            int a = 0;
            for (int k = 1; k < 3; ++k)
                if (cb.hi[k] - cb.lo[k] > cb.hi[a] - cb.lo[a]) a = k;
            mid = lo + n / 2;
            std::nth_element(prims.begin() + lo, prims.begin() + mid, prims.begin() + hi,
                             [a](const Prim &p, const Prim &q) { return p.c[a] < q.c[a]; });
        }
        if (mid == lo || mid == hi) mid = lo + n / 2;
        if (n <= max_leaf && (mid == lo || mid == hi)) return make_leaf(lo, hi, box);

        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        Box b0, b1;
        int32_t c0, c1;
        if (forks_left > 0 && n >= 16384) {
            std::vector<BvhNode> second;
            std::thread other([&] { c1 = build(mid, hi, depth + 1, b1, second, forks_left - 1); });
            c0 = build(lo, mid, depth + 1, b0, nodes, forks_left - 1);
            other.join();
            const int32_t shift = (int32_t)nodes.size();
            for (BvhNode &moved : second) {
                if (moved.child[0] >= 0) moved.child[0] += shift;
                if (moved.child[1] >= 0) moved.child[1] += shift;
            }
            if (c1 >= 0) c1 += shift;
            nodes.insert(nodes.end(), second.begin(), second.end());
        } else {
            c0 = build(lo, mid, depth + 1, b0, nodes, 0);
            c1 = build(mid, hi, depth + 1, b1, nodes, 0);
        }
        BvhNode &nd = nodes[me];
        nd.x[0] = b0.lo[0], nd.x[1] = b0.hi[0], nd.x[2] = b1.lo[0], nd.x[3] = b1.hi[0];
        nd.y[0] = b0.lo[1], nd.y[1] = b0.hi[1], nd.y[2] = b1.lo[1], nd.y[3] = b1.hi[1];
        nd.z[0] = b0.lo[2], nd.z[1] = b0.hi[2], nd.z[2] = b1.lo[2], nd.z[3] = b1.hi[2];
        nd.child[0] = c0, nd.child[1] = c1;
        nd.pad[0] = nd.pad[1] = 0;
        box = b0;
        box.grow(b1);
        return me;
    }
};

} // namespace

void build_flat_bvh(const MeshView &mesh, const std::vector<int32_t> &rank, int max_leaf_faces, int sah_bins, FlatBvh &out) {
    out = FlatBvh();
    if (max_leaf_faces <= 0) max_leaf_faces = 4;
    max_leaf_faces = std::min(max_leaf_faces, kMaxLeafFaces);
    if (sah_bins <= 0) sah_bins = 16;
    sah_bins = std::min(std::max(sah_bins, 4), 64);

    SahBuilder sb{mesh, rank, {}, out, max_leaf_faces, sah_bins};
    sb.prims.reserve((size_t)mesh.num_faces);
    for (int32_t f = 0; f < mesh.num_faces; ++f) {
        const float *n = mesh.face_normals + 3 * (size_t)f;
        const float nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        // a face without a usable plane can never be hit: every plane distance on it is NaN
        if (!(nn > 0.0f) || !std::isfinite(nn) || !std::isfinite(mesh.face_origin_distance[f])) continue;
        const int32_t *id = mesh.face_vertex_ids + 3 * (size_t)f;
        Prim p;
        p.box.reset();
        p.box.grow(vtx(mesh, id[0]));
        p.box.grow(vtx(mesh, id[1]));
        p.box.grow(vtx(mesh, id[2]));
        for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
        p.face = f;
        sb.prims.push_back(p);
    }
    out.num_device_faces = (int)sb.prims.size();
    out.tris.resize(sb.prims.size());
    out.nodes.reserve(sb.prims.size() / 2 + 2);

    const int n = (int)sb.prims.size();
    Box rb;
    int32_t root = 0;
    if (n > 0) root = sb.build(0, n, 0, rb, out.nodes, 4); // up to 16 threads at the fourth level
    out.num_leaves = sb.leaves.load();
    out.depth = sb.deepest.load();
    if (n == 0 || root < 0) {
        // The kernels start at inner node 0: wrap a lone leaf (or nothing) in one node whose second
        // child is a leaf with zero records.  Its inverted box does NOT keep rays out (the slab test
        // takes min/max of the two bounds), so the kernels skip zero-record leaves themselves.
        BvhNode nd;
        std::memset(&nd, 0, sizeof nd);
        const float inf = INFINITY;
        if (n > 0) {
            nd.x[0] = rb.lo[0], nd.x[1] = rb.hi[0];
            nd.y[0] = rb.lo[1], nd.y[1] = rb.hi[1];
            nd.z[0] = rb.lo[2], nd.z[1] = rb.hi[2];
            nd.child[0] = root;
        } else {
            nd.x[0] = nd.y[0] = nd.z[0] = inf, nd.x[1] = nd.y[1] = nd.z[1] = -inf;
            nd.child[0] = ~0;
        }
        nd.x[2] = nd.y[2] = nd.z[2] = inf, nd.x[3] = nd.y[3] = nd.z[3] = -inf;
        nd.child[1] = ~0; // leaf with zero records
        out.nodes.insert(out.nodes.begin(), nd);
        out.depth = std::max(out.depth, 1);
    }
    collapse_bvh(out, 4);
}

void gather_face_records(const MeshView &m, const std::vector<int32_t> &rank, std::vector<TriRecord> &out) {
    out.clear();
    out.reserve((size_t)m.num_faces);
    for (int32_t f = 0; f < m.num_faces; ++f) {
        const float *n = m.face_normals + 3 * (size_t)f;
        const float nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        if (!(nn > 0.0f) || !std::isfinite(nn) || !std::isfinite(m.face_origin_distance[f])) continue; // as build_flat_bvh
        const int32_t *id = m.face_vertex_ids + 3 * (size_t)f;
        TriRecord t;
        std::memset(&t, 0, sizeof t);
        std::memcpy(t.n, n, 12);
        t.d = m.face_origin_distance[f];
        std::memcpy(t.v0, vtx(m, id[0]), 12);
        std::memcpy(t.v1, vtx(m, id[1]), 12);
        std::memcpy(t.v2, vtx(m, id[2]), 12);
        t.face = f;
        t.rank = rank[f];
        out.push_back(t);
    }
}

namespace {

struct Collapser {
    FlatBvh &b;
    int max_children;

    struct Slot {
        int32_t ref;
        float lo[3], hi[3];
        float half_area() const {
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            return dx * dy + dy * dz + dz * dx;
        }
    };

    static Slot slot_of(const BvhNode &n, int k) {
        Slot s;
        s.ref = n.child[k];
        s.lo[0] = n.x[2 * k], s.hi[0] = n.x[2 * k + 1];
        s.lo[1] = n.y[2 * k], s.hi[1] = n.y[2 * k + 1];
        s.lo[2] = n.z[2 * k], s.hi[2] = n.z[2 * k + 1];
        return s;
    }

    // `held` = stack entries a walk may already hold when it arrives at this node
    int32_t run(int32_t n2, int depth, int held) {
        Slot c[4];
        int k = 2;
        c[0] = slot_of(b.nodes[(size_t)n2], 0);
        c[1] = slot_of(b.nodes[(size_t)n2], 1);
        while (k < max_children) {
            int pick = -1;
            float area = -1.0f;
            for (int i = 0; i < k; ++i)
                if (c[i].ref >= 0 && c[i].half_area() > area) area = c[i].half_area(), pick = i;
            if (pick < 0) break;
            const BvhNode &m = b.nodes[(size_t)c[pick].ref];
            c[pick] = slot_of(m, 0);
            c[k++] = slot_of(m, 1);
        }
        const int32_t me = (int32_t)b.nodes4.size();
        b.nodes4.emplace_back();
        b.depth4 = std::max(b.depth4, depth + 1);
        b.stack_need = std::max(b.stack_need, held + k - 1);
        int32_t refs[4];
        for (int i = 0; i < k; ++i) refs[i] = c[i].ref >= 0 ? run(c[i].ref, depth + 1, held + k - 1) : c[i].ref;
        Bvh4Node &nd = b.nodes4[(size_t)me];
        std::memset(&nd, 0, sizeof nd);
        nd.num_children = k;
        for (int i = 0; i < 4; ++i) {
            const bool used = i < k;
            nd.lox[i] = used ? c[i].lo[0] : FLT_MAX, nd.hix[i] = used ? c[i].hi[0] : -FLT_MAX;
            nd.loy[i] = used ? c[i].lo[1] : FLT_MAX, nd.hiy[i] = used ? c[i].hi[1] : -FLT_MAX;
            nd.loz[i] = used ? c[i].lo[2] : FLT_MAX, nd.hiz[i] = used ? c[i].hi[2] : -FLT_MAX;
            nd.child[i] = used ? refs[i] : ~0;
        }
        return me;
    }
};

} // namespace

void collapse_bvh(FlatBvh &bvh, int max_children) {
    bvh.nodes4.clear();
    bvh.depth4 = 0;
    bvh.stack_need = 0;
    bvh.nodes4.reserve(bvh.nodes.size() / 2 + 2);
    Collapser c{bvh, std::min(std::max(max_children, 2), 4)};
    c.run(0, 0, 0);
}

} // namespace rtamd
