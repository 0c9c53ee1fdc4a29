// Shadow-ray candidate grids.
//
// getLightIntensity passes the light sample's POSITION as the ray direction (src/flyscene.cpp:884), so all
// shadow rays of one sample are parallel: whether the sample is occluded at a point depends only on which
// faces lie "above" that point along the sample's direction.  Per sample the faces are therefore binned, once
// per scene, into a 2D grid on the plane perpendicular to that direction; a shadow ray is one cell lookup
// followed by the exact plane + three-edge test (src/flyscene.cpp:752-789) of the faces listed in the cell --
// no tree walk.  The lists are a conservative superset (projected bounding boxes, widened by far more than the
// rounding of the exact test), sorted by their farthest extent along the direction so that the scan stops at
// the first face lying wholly behind the ray's origin.  The exact test decides, so the result is the one the
// tree walk gives.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "bvh.hpp"

namespace rtamd {

// 64 B, one per light sample; read through scalar loads by the kernel.
struct ShadowGridHeader {
    float U[3], V[3], W[3]; // orthonormal: W = direction / |direction|
    float u0, v0;           // grid origin on the (U, V) plane
    float inv_cell;         // cells per unit length
    int32_t res;            // cells per side
    int32_t first_cell;     // index of cell (0,0) in the cell-start array (each grid holds res*res + 1 entries)
    int32_t pad[2];
};
static_assert(sizeof(ShadowGridHeader) == 64, "ShadowGridHeader layout");

struct ShadowGridEntry {
    int32_t tri; // TriRecord index
    float zmax;  // farthest extent of the face along W, widened; a ray origin beyond it cannot see the face
};

struct ShadowGrids {
    bool usable = false; // false: some direction is degenerate or the lists would be too large -> tree walk
    std::vector<ShadowGridHeader> headers;
    std::vector<int32_t> cell_start; // absolute indices into `entries`
    std::vector<ShadowGridEntry> entries;
};

// The same grids built on the GPU (grid_build.hip): arrays stay in device memory, owned by the caller afterwards.
struct DeviceShadowGrids {
    bool usable = false;
    std::vector<ShadowGridHeader> headers;
    void *d_cell_start = nullptr; // int32, absolute indices into the entries
    void *d_entries = nullptr;    // ShadowGridEntry
    uint64_t num_cells = 0, num_entries = 0;
};
// tris: host copy of the records d_tris holds on the device.  false (with a reason): build on the host instead.
bool build_shadow_grids_device(const std::vector<TriRecord> &tris, const void *d_tris, const float *samples, int num_samples, double density,
                               DeviceShadowGrids &out, std::string &why);
double shadow_grid_density(); // cells per face

// samples = num_samples * 3 (the light sample positions = shadow ray directions).
void build_shadow_grids(const FlatBvh &bvh, const float *samples, int num_samples, ShadowGrids &out);

} // namespace rtamd
