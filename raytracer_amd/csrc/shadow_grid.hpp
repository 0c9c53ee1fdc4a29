// Shadow-ray candidate grids.
//
// getLightIntensity passes the light sample's POSITION as the ray direction (src/flyscene.cpp:884), so all
// shadow rays of one sample are parallel: whether the sample is occluded at a point depends only on which
// faces lie "above" that point along the sample's direction.  Per sample the faces are therefore binned, once
// per scene, into a 2D grid on the plane perpendicular to that direction; a shadow ray is one cell lookup
// followed by the exact plane + three-edge test (src/flyscene.cpp:752-789) of the faces listed in the cell --
// no tree walk.  The lists are a conservative superset (projected bounding boxes, less the cells that the projected
// triangle cannot touch, all widened by far more than the rounding of the exact test), sorted by their farthest extent along the direction so that the scan stops at
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

// Entries of padding behind the last list of the entry array: the kernel fetches its batch of entries (three) with one
// address, whatever the length of the list, and must find memory there.
constexpr int kShadowGridPadding = 7;

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
double shadow_grid_density(size_t num_faces); // cells per face for a mesh of that many faces

// ---- which cells of its box a face really needs (shared by the host and the device builder: same lists) ----
// The box of a triangle holds about twice the cells the triangle touches.  A cell is left out when its rectangle,
// widened by `slack`, lies wholly outside one edge of the projected triangle.  Double precision, the same
// expressions on both sides.  Not applied (on = false) to faces whose box is one cell wide or high, or holds more
// than kCullMaxCells cells (a ground plane: millions), nor to projected slivers, whose orientation is not safe to
// take from the sign of a rounded area.
#ifdef __HIP__
#define RTAMD_HD __host__ __device__
#else
#define RTAMD_HD
#endif
constexpr int64_t kCullMaxCells = 2048;

struct CellCull {
    bool on;
    double ex[3], ey[3], dx[3], dy[3]; // edge k: from (ex, ey) along (dx, dy), the inside on its left
};

RTAMD_HD inline CellCull make_cell_cull(const float *v0, const float *v1, const float *v2, const float *U, const float *V, int64_t box_w,
                                        int64_t box_h, double margin) {
    CellCull c;
    c.on = false;
    if (box_w < 2 || box_h < 2 || box_w * box_h > kCullMaxCells) return c;
    const float *vs[3] = {v0, v1, v2};
    double pu[3], pv[3];
    for (int k = 0; k < 3; ++k) {
        pu[k] = (double)vs[k][0] * U[0] + (double)vs[k][1] * U[1] + (double)vs[k][2] * U[2];
        pv[k] = (double)vs[k][0] * V[0] + (double)vs[k][1] * V[1] + (double)vs[k][2] * V[2];
    }
    const double area = (pu[1] - pu[0]) * (pv[2] - pv[0]) - (pv[1] - pv[0]) * (pu[2] - pu[0]);
    double around = 0.0;
    for (int k = 0; k < 3; ++k) {
        const int n = k == 2 ? 0 : k + 1;
        const double du = pu[n] - pu[k], dv = pv[n] - pv[k];
        around += (du < 0 ? -du : du) + (dv < 0 ? -dv : dv);
    }
    const double mag = area < 0 ? -area : area;
    if (!(mag > 1e-6 * around * around) || !(mag > 4.0 * margin * around)) return c; // a sliver, or thinner than the margin
    c.on = true;
    for (int k = 0; k < 3; ++k) {
        const int n = k == 2 ? 0 : k + 1;
        const int from = area > 0 ? k : n, to = area > 0 ? n : k;
        c.ex[k] = pu[from], c.ey[k] = pv[from];
        c.dx[k] = pu[to] - pu[from], c.dy[k] = pv[to] - pv[from];
    }
    return c;
}

// u0, v0: grid origin; cell: edge length of a cell; slack: widening of the cell's rectangle (the builder's margin plus
// the rounding of the kernel's cell lookup).  E(r) = dx (r.y - ey) - dy (r.x - ex) is >= 0 on the inside of an edge.
RTAMD_HD inline bool cell_touched(const CellCull &c, int x, int y, double u0, double v0, double cell, double slack) {
    if (!c.on) return true;
    const double rx0 = u0 + x * cell - slack, rx1 = u0 + (x + 1) * cell + slack;
    const double ry0 = v0 + y * cell - slack, ry1 = v0 + (y + 1) * cell + slack;
    for (int k = 0; k < 3; ++k) {
        const double cy = c.dx[k] >= 0 ? ry1 : ry0;
        const double cx = c.dy[k] >= 0 ? rx0 : rx1;
        if (c.dx[k] * (cy - c.ey[k]) - c.dy[k] * (cx - c.ex[k]) < 0) return false;
    }
    return true;
}

// slack for a grid: the builder's margin and four thousandths of a cell (the kernel finds the cell with two float
// operations on values up to 4096 cells: 5e-4 cell of rounding at most)
RTAMD_HD inline double cell_slack(double margin, double cell) { return margin + 4e-3 * cell; }

// samples = num_samples * 3 (the light sample positions = shadow ray directions).
void build_shadow_grids(const FlatBvh &bvh, const float *samples, int num_samples, ShadowGrids &out);

} // namespace rtamd
