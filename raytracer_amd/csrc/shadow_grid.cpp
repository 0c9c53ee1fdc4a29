#include "shadow_grid.hpp"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace rtamd {

namespace {

constexpr int kMaxSamplesWithGrids = 64;
constexpr int64_t kMaxEntriesPerGrid = 1ll << 27; // 1 GiB of entries per sample at most
constexpr double kMarginRel = 3e-5;              // of the largest coordinate magnitude; the exact test's own rounding is ~1e-7

struct OneGrid {
    ShadowGridHeader h{};
    std::vector<int32_t> start; // res*res + 1, relative
    std::vector<ShadowGridEntry> entries;
    bool ok = false;
};

// Runs fn(first, last) over [0, n) on `threads` threads (the calling one included).
template <class F>
void parallel_ranges(size_t n, int threads, F fn) {
    if (threads <= 1 || n < 4096) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(fn, n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads);
    fn((size_t)0, n / (size_t)threads);
    for (std::thread &t : pool) t.join();
}

// One grid, built by `threads` threads: projection, counting and filling run over ranges of faces (atomic
// per-cell counters), the sort of the lists over ranges of cells; the sort makes the result independent of the
// order in which the threads filled the lists.
void build_one(const FlatBvh &bvh, const float *dir, double scale, OneGrid &g, int threads) {
    const double dx = dir[0], dy = dir[1], dz = dir[2];
    const double len2 = dx * dx + dy * dy + dz * dz;
    if (!std::isfinite(len2) || !(len2 > 1e-60)) return;
    const double len = std::sqrt(len2);
    const double W[3] = {dx / len, dy / len, dz / len};
    // any unit vector not parallel to W
    int small = 0;
    if (std::fabs(W[1]) < std::fabs(W[small])) small = 1;
    if (std::fabs(W[2]) < std::fabs(W[small])) small = 2;
    double a[3] = {0, 0, 0};
    a[small] = 1.0;
    double U[3] = {a[1] * W[2] - a[2] * W[1], a[2] * W[0] - a[0] * W[2], a[0] * W[1] - a[1] * W[0]};
    const double ul = std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
    for (double &c : U) c /= ul;
    const double V[3] = {W[1] * U[2] - W[2] * U[1], W[2] * U[0] - W[0] * U[2], W[0] * U[1] - W[1] * U[0]};
    // the kernel projects with these float values; project the faces with the same values
    float Uf[3], Vf[3], Wf[3];
    for (int k = 0; k < 3; ++k) Uf[k] = (float)U[k], Vf[k] = (float)V[k], Wf[k] = (float)W[k];

    const size_t n = bvh.tris.size();
    const double margin = kMarginRel * scale;
    struct Box2 {
        float ulo, uhi, vlo, vhi, zmax;
    };
    std::vector<Box2> box(n);
    struct Extent {
        double u0 = DBL_MAX, u1 = -DBL_MAX, v0 = DBL_MAX, v1 = -DBL_MAX;
        bool finite = true;
    };
    std::vector<Extent> part((size_t)std::max(threads, 1));
    std::atomic<int> next_part{0};
    parallel_ranges(n, threads, [&](size_t first, size_t last) {
        Extent e;
        for (size_t i = first; i < last; ++i) {
            const TriRecord &t = bvh.tris[i];
            const float *vs[3] = {t.v0, t.v1, t.v2};
            double ulo = DBL_MAX, uhi = -DBL_MAX, vlo = DBL_MAX, vhi = -DBL_MAX, zhi = -DBL_MAX;
            for (const float *p : vs) {
                const double u = (double)p[0] * Uf[0] + (double)p[1] * Uf[1] + (double)p[2] * Uf[2];
                const double v = (double)p[0] * Vf[0] + (double)p[1] * Vf[1] + (double)p[2] * Vf[2];
                const double z = (double)p[0] * Wf[0] + (double)p[1] * Wf[1] + (double)p[2] * Wf[2];
                ulo = std::min(ulo, u), uhi = std::max(uhi, u), vlo = std::min(vlo, v), vhi = std::max(vhi, v), zhi = std::max(zhi, z);
            }
            if (!std::isfinite(ulo + uhi + vlo + vhi + zhi)) {
                e.finite = false;
                break;
            }
            Box2 &b = box[i];
            b = Box2{(float)(ulo - margin), (float)(uhi + margin), (float)(vlo - margin), (float)(vhi + margin), (float)(zhi + margin)};
            b.ulo = std::nextafterf(b.ulo, -FLT_MAX), b.uhi = std::nextafterf(b.uhi, FLT_MAX);
            b.vlo = std::nextafterf(b.vlo, -FLT_MAX), b.vhi = std::nextafterf(b.vhi, FLT_MAX);
            b.zmax = std::nextafterf(b.zmax, FLT_MAX);
            e.u0 = std::min(e.u0, (double)b.ulo), e.u1 = std::max(e.u1, (double)b.uhi);
            e.v0 = std::min(e.v0, (double)b.vlo), e.v1 = std::max(e.v1, (double)b.vhi);
        }
        part[(size_t)next_part.fetch_add(1)] = e;
    });
    double gu0 = DBL_MAX, gu1 = -DBL_MAX, gv0 = DBL_MAX, gv1 = -DBL_MAX;
    for (int k = 0; k < next_part.load(); ++k) {
        const Extent &e = part[(size_t)k];
        if (!e.finite) return;
        gu0 = std::min(gu0, e.u0), gu1 = std::max(gu1, e.u1), gv0 = std::min(gv0, e.v0), gv1 = std::max(gv1, e.v1);
    }
    if (n == 0) gu0 = gv0 = 0.0, gu1 = gv1 = 1.0;
    const double extent = std::max(std::max(gu1 - gu0, gv1 - gv0), 1e-30);
    // Cells per face.  The kernel is bound by fetching the lists, and finer cells mean shorter lists (fewer faces
    // that merely come near the cell): 4 cells per face against 2 take 6-10 % off the shadow rays for 1.7x the
    // entries, 6 another 2-4 % for 1.3x; beyond that see shadow_grid_density.
    const double density = shadow_grid_density(n);
    int res = (int)std::lround(std::sqrt(density * (double)std::max<size_t>(n, 1)));
    res = std::min(std::max(res, 16), 4096);

    for (;; res /= 2) { // halve the resolution until the lists fit
        const float inv_cell = (float)((double)res / (extent * (1.0 + 1e-6)));
        const float u0 = (float)gu0, v0 = (float)gv0;
        auto cell = [&](float x, float origin) {
            // same expression as the kernel; the widened boxes absorb its rounding
            const float f = std::floor((x - origin) * inv_cell);
            return (int)std::min(std::max(f, 0.0f), (float)(res - 1));
        };
        const size_t ncells = (size_t)res * res;
        const double cell_size = 1.0 / (double)inv_cell, slack = cell_slack(margin, cell_size);
        auto cull_of = [&](size_t i, int x0, int x1, int y0, int y1) {
            const TriRecord &t = bvh.tris[i];
            return make_cell_cull(t.v0, t.v1, t.v2, Uf, Vf, (int64_t)x1 - x0 + 1, (int64_t)y1 - y0 + 1, margin);
        };
        std::vector<std::atomic<int32_t>> count(ncells);
        for (std::atomic<int32_t> &c : count) c.store(0, std::memory_order_relaxed);
        std::atomic<int64_t> total{0};
        parallel_ranges(n, threads, [&](size_t first, size_t last) {
            int64_t mine = 0;
            for (size_t i = first; i < last && total.load(std::memory_order_relaxed) <= kMaxEntriesPerGrid; ++i) {
                const int x0 = cell(box[i].ulo, u0), x1 = cell(box[i].uhi, u0), y0 = cell(box[i].vlo, v0), y1 = cell(box[i].vhi, v0);
                if ((int64_t)(x1 - x0 + 1) * (y1 - y0 + 1) > kMaxEntriesPerGrid) {
                    mine = kMaxEntriesPerGrid + 1;
                    break;
                }
                const CellCull cull = cull_of(i, x0, x1, y0, y1);
                for (int y = y0; y <= y1; ++y)
                    for (int x = x0; x <= x1; ++x)
                        if (cell_touched(cull, x, y, (double)u0, (double)v0, cell_size, slack)) {
                            count[(size_t)y * res + x].fetch_add(1, std::memory_order_relaxed);
                            ++mine;
                        }
                if (mine > kMaxEntriesPerGrid) break;
            }
            total.fetch_add(mine);
        });
        if (total.load() > kMaxEntriesPerGrid) {
            if (res <= 16) return;
            continue;
        }
        g.start.assign(ncells + 1, 0);
        int32_t run = 0;
        for (size_t c = 0; c < ncells; ++c) {
            g.start[c] = run;
            run += count[c].load(std::memory_order_relaxed);
            count[c].store(g.start[c], std::memory_order_relaxed); // from here on: the fill cursor of the cell
        }
        g.start[ncells] = run;
        g.entries.resize((size_t)run);
        parallel_ranges(n, threads, [&](size_t first, size_t last) {
            for (size_t i = first; i < last; ++i) {
                const int x0 = cell(box[i].ulo, u0), x1 = cell(box[i].uhi, u0), y0 = cell(box[i].vlo, v0), y1 = cell(box[i].vhi, v0);
                const CellCull cull = cull_of(i, x0, x1, y0, y1);
                for (int y = y0; y <= y1; ++y)
                    for (int x = x0; x <= x1; ++x)
                        if (cell_touched(cull, x, y, (double)u0, (double)v0, cell_size, slack))
                            g.entries[(size_t)count[(size_t)y * res + x].fetch_add(1, std::memory_order_relaxed)] = ShadowGridEntry{(int32_t)i, box[i].zmax};
            }
        });
        parallel_ranges(ncells, threads, [&](size_t first, size_t last) {
            for (size_t c = first; c < last; ++c)
                std::sort(g.entries.begin() + g.start[c], g.entries.begin() + g.start[c + 1],
                          [](const ShadowGridEntry &p, const ShadowGridEntry &q) { return p.zmax > q.zmax || (p.zmax == q.zmax && p.tri < q.tri); });
        });
        std::memcpy(g.h.U, Uf, 12), std::memcpy(g.h.V, Vf, 12), std::memcpy(g.h.W, Wf, 12);
        g.h.u0 = u0, g.h.v0 = v0, g.h.inv_cell = inv_cell, g.h.res = res;
        g.ok = true;
        return;
    }
}

} // namespace

double shadow_grid_density(size_t num_faces) {
    // Nine cells per face for a mesh of up to half a million faces, six beyond: the shadow kernel waits for round trips to
    // memory, and shorter lists are fewer trips (69 K-face bunny: frames 1.6-2 % faster, a frame alone 4 %, for 1.3x the
    // entries); the entries of a 1.1 M-face mesh (110 MB per grid at nine) no longer pay for it (1.6 % slower).
    static const double forced = std::getenv("RTAMD_GRID_DENSITY") ? std::atof(std::getenv("RTAMD_GRID_DENSITY")) : 0.0;
    return forced > 0.0 ? forced : num_faces <= ((size_t)1 << 19) ? 9.0 : 6.0;
}

void build_shadow_grids(const FlatBvh &bvh, const float *samples, int num_samples, ShadowGrids &out) {
    out = ShadowGrids();
    if (num_samples <= 0 || num_samples > kMaxSamplesWithGrids) return;
    double scale = 1e-30;
    for (const TriRecord &t : bvh.tris)
        for (int k = 0; k < 3; ++k) scale = std::max(scale, (double)std::max(std::fabs(t.v0[k]), std::max(std::fabs(t.v1[k]), std::fabs(t.v2[k]))));
    if (!std::isfinite(scale)) return;

    std::vector<OneGrid> grids((size_t)num_samples);
    // up to 16 threads in all: one per sample, and the rest inside the samples when those are few
    const int budget = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
    const int workers = std::min(num_samples, budget);
    const int inner = std::max(1, budget / workers);
    std::vector<std::thread> pool;
    for (int w = 0; w < workers; ++w)
        pool.emplace_back([&, w] {
            for (int k = w; k < num_samples; k += workers) build_one(bvh, samples + 3 * (size_t)k, scale, grids[(size_t)k], inner);
        });
    for (std::thread &t : pool) t.join();

    int64_t cells = 0, entries = 0;
    for (const OneGrid &g : grids) {
        if (!g.ok) return;
        cells += (int64_t)g.start.size();
        entries += (int64_t)g.entries.size();
    }
    if (cells >= (1ll << 31) || entries >= (1ll << 31)) return;
    out.cell_start.reserve((size_t)cells);
    out.entries.reserve((size_t)entries);
    for (OneGrid &g : grids) {
        g.h.first_cell = (int32_t)out.cell_start.size();
        const int32_t base = (int32_t)out.entries.size();
        for (int32_t s : g.start) out.cell_start.push_back(base + s);
        out.entries.insert(out.entries.end(), g.entries.begin(), g.entries.end());
        out.headers.push_back(g.h);
        std::vector<int32_t>().swap(g.start);
        std::vector<ShadowGridEntry>().swap(g.entries);
    }
    out.usable = true;
}

} // namespace rtamd
