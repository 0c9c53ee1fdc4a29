// On-GPU construction of the four-way BVH (optional builder, rtamd_build_params::builder).
//
// Stands in for the host SAH builder (bvh.cpp) where time to first pixel matters more than the last bit of
// tree quality: Morton codes of the faces' centres, a radix sort, the binary radix tree of Karras ("Maximizing
// parallelism in the construction of BVHs, octrees, and k-d trees", 2012), a bottom-up fit of the boxes, and a
// collapse of every node with its two children into one four-way node.  Like every hierarchy here it only
// proposes candidates; which face a ray hits is decided by the exact test and the faces' ranks.
#pragma once
#include <string>

#include "bvh.hpp"

namespace rtamd {

// `faces`: TriRecords in any order (face and rank filled in).  On success fills out.tris (sorted), out.nodes4,
// out.num_leaves, out.depth4, out.stack_need and returns true; returns false (with a message) when the builder
// does not apply (fewer than two faces, tree too deep for the traversal stack, HIP error): the caller then
// falls back to the host builder.
bool build_lbvh_device(const std::vector<TriRecord> &faces, int max_leaf_faces, FlatBvh &out, std::string &why);

} // namespace rtamd
