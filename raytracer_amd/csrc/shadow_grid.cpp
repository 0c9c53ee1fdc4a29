#include "shadow_grid.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
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

void build_one(const FlatBvh &bvh, const float *dir, double scale, OneGrid &g) {
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
    double gu0 = DBL_MAX, gu1 = -DBL_MAX, gv0 = DBL_MAX, gv1 = -DBL_MAX;
    for (size_t i = 0; i < n; ++i) {
        const TriRecord &t = bvh.tris[i];
        const float *vs[3] = {t.v0, t.v1, t.v2};
        double ulo = DBL_MAX, uhi = -DBL_MAX, vlo = DBL_MAX, vhi = -DBL_MAX, zhi = -DBL_MAX;
        for (const float *p : vs) {
            const double u = (double)p[0] * Uf[0] + (double)p[1] * Uf[1] + (double)p[2] * Uf[2];
            const double v = (double)p[0] * Vf[0] + (double)p[1] * Vf[1] + (double)p[2] * Vf[2];
            const double z = (double)p[0] * Wf[0] + (double)p[1] * Wf[1] + (double)p[2] * Wf[2];
            ulo = std::min(ulo, u), uhi = std::max(uhi, u), vlo = std::min(vlo, v), vhi = std::max(vhi, v), zhi = std::max(zhi, z);
        }
        if (!std::isfinite(ulo + uhi + vlo + vhi + zhi)) return;
        box[i] = Box2{(float)(ulo - margin), (float)(uhi + margin), (float)(vlo - margin), (float)(vhi + margin), (float)(zhi + margin)};
        box[i].ulo = std::nextafterf(box[i].ulo, -FLT_MAX), box[i].uhi = std::nextafterf(box[i].uhi, FLT_MAX);
        box[i].vlo = std::nextafterf(box[i].vlo, -FLT_MAX), box[i].vhi = std::nextafterf(box[i].vhi, FLT_MAX);
        box[i].zmax = std::nextafterf(box[i].zmax, FLT_MAX);
        gu0 = std::min(gu0, (double)box[i].ulo), gu1 = std::max(gu1, (double)box[i].uhi);
        gv0 = std::min(gv0, (double)box[i].vlo), gv1 = std::max(gv1, (double)box[i].vhi);
    }
    if (n == 0) gu0 = gv0 = 0.0, gu1 = gv1 = 1.0;
    const double extent = std::max(std::max(gu1 - gu0, gv1 - gv0), 1e-30);
    int res = (int)std::lround(std::sqrt(2.0 * (double)std::max<size_t>(n, 1)));
    res = std::min(std::max(res, 16), 4096);

    for (;; res /= 2) { // halve the resolution until the lists fit
        const float inv_cell = (float)((double)res / (extent * (1.0 + 1e-6)));
        const float u0 = (float)gu0, v0 = (float)gv0;
        auto cell = [&](float x, float origin) {
            // same expression as the kernel; the widened boxes absorb its rounding
            const float f = std::floor((x - origin) * inv_cell);
            return (int)std::min(std::max(f, 0.0f), (float)(res - 1));
        };
        std::vector<int32_t> count((size_t)res * res + 1, 0);
        int64_t total = 0;
        for (size_t i = 0; i < n; ++i) {
            const int x0 = cell(box[i].ulo, u0), x1 = cell(box[i].uhi, u0), y0 = cell(box[i].vlo, v0), y1 = cell(box[i].vhi, v0);
            total += (int64_t)(x1 - x0 + 1) * (y1 - y0 + 1);
            if (total > kMaxEntriesPerGrid) break;
            for (int y = y0; y <= y1; ++y)
                for (int x = x0; x <= x1; ++x) ++count[(size_t)y * res + x];
        }
        if (total > kMaxEntriesPerGrid) {
            if (res <= 16) return;
            continue;
        }
        g.start.assign((size_t)res * res + 1, 0);
        int32_t run = 0;
        for (size_t c = 0; c < (size_t)res * res; ++c) {
            g.start[c] = run;
            run += count[c];
        }
        g.start[(size_t)res * res] = run;
        g.entries.resize((size_t)run);
        std::vector<int32_t> fill(g.start.begin(), g.start.end() - 1);
        for (size_t i = 0; i < n; ++i) {
            const int x0 = cell(box[i].ulo, u0), x1 = cell(box[i].uhi, u0), y0 = cell(box[i].vlo, v0), y1 = cell(box[i].vhi, v0);
            for (int y = y0; y <= y1; ++y)
                for (int x = x0; x <= x1; ++x) g.entries[(size_t)fill[(size_t)y * res + x]++] = ShadowGridEntry{(int32_t)i, box[i].zmax};
        }
        for (size_t c = 0; c < (size_t)res * res; ++c)
            std::sort(g.entries.begin() + g.start[c], g.entries.begin() + g.start[c + 1],
                      [](const ShadowGridEntry &p, const ShadowGridEntry &q) { return p.zmax > q.zmax || (p.zmax == q.zmax && p.tri < q.tri); });
        std::memcpy(g.h.U, Uf, 12), std::memcpy(g.h.V, Vf, 12), std::memcpy(g.h.W, Wf, 12);
        g.h.u0 = u0, g.h.v0 = v0, g.h.inv_cell = inv_cell, g.h.res = res;
        g.ok = true;
        return;
    }
}

} // namespace

void build_shadow_grids(const FlatBvh &bvh, const float *samples, int num_samples, ShadowGrids &out) {
    out = ShadowGrids();
    if (num_samples <= 0 || num_samples > kMaxSamplesWithGrids) return;
    double scale = 1e-30;
    for (const TriRecord &t : bvh.tris)
        for (int k = 0; k < 3; ++k) scale = std::max(scale, (double)std::max(std::fabs(t.v0[k]), std::max(std::fabs(t.v1[k]), std::fabs(t.v2[k]))));
    if (!std::isfinite(scale)) return;

    std::vector<OneGrid> grids((size_t)num_samples);
    const int workers = std::max(1, std::min(num_samples, std::min(16, (int)std::thread::hardware_concurrency())));
    std::vector<std::thread> pool;
    for (int w = 0; w < workers; ++w)
        pool.emplace_back([&, w] {
            for (int k = w; k < num_samples; k += workers) build_one(bvh, samples + 3 * (size_t)k, scale, grids[(size_t)k]);
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
