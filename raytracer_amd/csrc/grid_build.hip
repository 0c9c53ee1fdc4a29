// On-GPU construction of the shadow-ray candidate grids (see shadow_grid.hpp), the counterpart of the threaded
// host builder in shadow_grid.cpp for scenes whose hierarchy is built on the device as well.  Per light sample:
// project every face (same double-precision expressions as the host builder, so the boxes are the same bits),
// reduce the extent, count the cells each box touches, emit (cell, far extent | face) keys, radix-sort them --
// ascending cell, descending far extent, and, the sort being stable, ascending face among equal extents, which is
// the order the host builder sorts its lists into -- and find the start of every cell by binary search.
#include "shadow_grid.hpp"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace rtamd {

namespace {

constexpr int64_t kMaxEntriesPerGridDevice = 1ll << 27;
constexpr double kMarginRelDevice = 3e-5; // as the host builder

struct Basis {
    float U[3], V[3], W[3];
};

// monotonic map float -> unsigned (larger float, larger key)
__device__ __forceinline__ unsigned orderable(float x) {
    const unsigned b = (unsigned)__float_as_int(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(unsigned k) {
    const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __int_as_float((int)b);
}

__global__ void project_kernel(const TriRecord *tris, int n, Basis bs, double margin, float *ulo, float *uhi, float *vlo, float *vhi, float *zmax,
                               int *not_finite) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const TriRecord &t = tris[i];
    const float *vs[3] = {t.v0, t.v1, t.v2};
    double lo_u = DBL_MAX, hi_u = -DBL_MAX, lo_v = DBL_MAX, hi_v = -DBL_MAX, hi_z = -DBL_MAX;
    for (int k = 0; k < 3; ++k) {
        const float *p = vs[k];
        const double u = (double)p[0] * bs.U[0] + (double)p[1] * bs.U[1] + (double)p[2] * bs.U[2];
        const double v = (double)p[0] * bs.V[0] + (double)p[1] * bs.V[1] + (double)p[2] * bs.V[2];
        const double z = (double)p[0] * bs.W[0] + (double)p[1] * bs.W[1] + (double)p[2] * bs.W[2];
        lo_u = fmin(lo_u, u), hi_u = fmax(hi_u, u), lo_v = fmin(lo_v, v), hi_v = fmax(hi_v, v), hi_z = fmax(hi_z, z);
    }
    if (!isfinite(lo_u + hi_u + lo_v + hi_v + hi_z)) {
        atomicOr(not_finite, 1);
        lo_u = hi_u = lo_v = hi_v = hi_z = 0.0;
    }
    ulo[i] = nextafterf((float)(lo_u - margin), -FLT_MAX), uhi[i] = nextafterf((float)(hi_u + margin), FLT_MAX);
    vlo[i] = nextafterf((float)(lo_v - margin), -FLT_MAX), vhi[i] = nextafterf((float)(hi_v + margin), FLT_MAX);
    zmax[i] = nextafterf((float)(hi_z + margin), FLT_MAX);
}

struct GridFrame {
    float u0, v0, inv_cell;
    int res;
};

__device__ __forceinline__ int cell_of(float x, float origin, const GridFrame &g) {
    const float f = floorf((x - origin) * g.inv_cell); // the expression of the shadow kernel and of the host builder
    return (int)fminf(fmaxf(f, 0.0f), (float)(g.res - 1));
}

// the cells of its box a face is entered into (shadow_grid.hpp: cell_touched)
struct CullFrame {
    Basis bs;
    double margin, cell, slack;
};

__global__ void count_kernel(const TriRecord *tris, int n, GridFrame g, CullFrame cf, const float *ulo, const float *uhi, const float *vlo,
                             const float *vhi, unsigned *counts) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int x0 = cell_of(ulo[i], g.u0, g), x1 = cell_of(uhi[i], g.u0, g), y0 = cell_of(vlo[i], g.v0, g), y1 = cell_of(vhi[i], g.v0, g);
    const CellCull cull = make_cell_cull(tris[i].v0, tris[i].v1, tris[i].v2, cf.bs.U, cf.bs.V, (int64_t)x1 - x0 + 1, (int64_t)y1 - y0 + 1, cf.margin);
    if (!cull.on) {
        counts[i] = (unsigned)(x1 - x0 + 1) * (unsigned)(y1 - y0 + 1);
        return;
    }
    unsigned c = 0;
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) c += cell_touched(cull, x, y, (double)g.u0, (double)g.v0, cf.cell, cf.slack) ? 1u : 0u;
    counts[i] = c;
}

constexpr unsigned kWideFace = 2048; // cells; a face touching more is emitted by a whole workgroup (a ground plane touches millions)

__global__ void emit_kernel(const TriRecord *tris, int n, GridFrame g, CullFrame cf, const float *ulo, const float *uhi, const float *vlo,
                            const float *vhi, const float *zmax, const unsigned *counts, const unsigned *offsets, unsigned long long *keys, int *vals) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n || counts[i] > kWideFace) return;
    const int x0 = cell_of(ulo[i], g.u0, g), x1 = cell_of(uhi[i], g.u0, g), y0 = cell_of(vlo[i], g.v0, g), y1 = cell_of(vhi[i], g.v0, g);
    const CellCull cull = make_cell_cull(tris[i].v0, tris[i].v1, tris[i].v2, cf.bs.U, cf.bs.V, (int64_t)x1 - x0 + 1, (int64_t)y1 - y0 + 1, cf.margin);
    const unsigned long long far_first = (unsigned long long)(~orderable(zmax[i])); // descending extent = ascending key
    size_t at = offsets[i];
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            if (!cell_touched(cull, x, y, (double)g.u0, (double)g.v0, cf.cell, cf.slack)) continue;
            keys[at] = ((unsigned long long)((unsigned)y * (unsigned)g.res + (unsigned)x) << 32) | far_first;
            vals[at] = i;
            ++at;
        }
}

// one workgroup per wide face (more than kWideFace = kCullMaxCells cells: its whole box is entered)
static_assert(kWideFace == kCullMaxCells, "a face wide enough for emit_wide_kernel must be one the cell test leaves alone");
__global__ void emit_wide_kernel(const int *wide, GridFrame g, const float *ulo, const float *uhi, const float *vlo, const float *vhi, const float *zmax,
                                 const unsigned *offsets, unsigned long long *keys, int *vals) {
    const int i = wide[blockIdx.x];
    const int x0 = cell_of(ulo[i], g.u0, g), x1 = cell_of(uhi[i], g.u0, g), y0 = cell_of(vlo[i], g.v0, g), y1 = cell_of(vhi[i], g.v0, g);
    const unsigned long long far_first = (unsigned long long)(~orderable(zmax[i]));
    const unsigned w = (unsigned)(x1 - x0 + 1), count = w * (unsigned)(y1 - y0 + 1);
    const size_t base = offsets[i];
    for (unsigned j = threadIdx.x; j < count; j += blockDim.x) {
        const unsigned y = (unsigned)y0 + j / w, x = (unsigned)x0 + j % w;
        keys[base + j] = ((unsigned long long)(y * (unsigned)g.res + x) << 32) | far_first;
        vals[base + j] = i;
    }
}

__global__ void entries_kernel(long long total, const unsigned long long *keys, const int *vals, int2 *entries) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    entries[j] = make_int2(vals[j], __float_as_int(from_orderable(~(unsigned)(keys[j] & 0xffffffffull))));
}

// start[c] = base + index of the first key of cell c (or of any later cell), c = 0 .. cells (inclusive)
__global__ void cell_start_kernel(long long total, int cells, const unsigned long long *keys, int base, int *start) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c > cells) return;
    long long lo = 0, hi = total;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((int)(keys[mid] >> 32) < c) lo = mid + 1; else hi = mid;
    }
    start[c] = base + (int)lo;
}

struct Buffers {
    std::vector<void *> all;
    ~Buffers() {
        for (void *p : all)
            if (p) (void)hipFree(p);
    }
    template <class T>
    bool get(T **p, size_t count) {
        void *raw = nullptr;
        if (hipMalloc(&raw, count * sizeof(T) + 256) != hipSuccess) return false;
        all.push_back(raw);
        *p = (T *)raw;
        return true;
    }
    void release(void *p) { // hand a buffer over to the caller
        for (void *&q : all)
            if (q == p) q = nullptr;
    }
};

} // namespace

bool build_shadow_grids_device(const std::vector<TriRecord> &tris, const void *d_tris, const float *samples, int num_samples, double density,
                               DeviceShadowGrids &out, std::string &why) {
    out = DeviceShadowGrids();
    const int n = (int)tris.size();
    if (num_samples <= 0 || num_samples > 64 || n == 0) {
        why = "no faces, or no / too many light samples";
        return false;
    }
    double scale = 1e-30;
    for (const TriRecord &t : tris)
        for (int k = 0; k < 3; ++k) scale = std::max(scale, (double)std::max(std::fabs(t.v0[k]), std::max(std::fabs(t.v1[k]), std::fabs(t.v2[k]))));
    if (!std::isfinite(scale)) {
        why = "non-finite vertex";
        return false;
    }
#define GRID_TRY(expr)                                                    \
    do {                                                                  \
        const hipError_t e_ = (expr);                                     \
        if (e_ != hipSuccess) {                                           \
            why = std::string(#expr " failed: ") + hipGetErrorString(e_); \
            return false;                                                 \
        }                                                                 \
    } while (0)

    const bool verbose = std::getenv("RTAMD_VERBOSE_BUILD") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double ms_project = 0, ms_count = 0, ms_alloc = 0, ms_sort = 0, ms_finish = 0, ms_free = 0, ms_concat = 0;
    Buffers buf;
    float *ulo, *uhi, *vlo, *vhi, *zmax, *d_extent;
    unsigned *counts, *offsets;
    int *d_flag;
    const size_t N = (size_t)n;
    if (!buf.get(&ulo, N) || !buf.get(&uhi, N) || !buf.get(&vlo, N) || !buf.get(&vhi, N) || !buf.get(&zmax, N) || !buf.get(&counts, N + 1) ||
        !buf.get(&offsets, N + 1) || !buf.get(&d_extent, 4) || !buf.get(&d_flag, 1)) {
        why = "out of device memory";
        return false;
    }
    size_t reduce_bytes = 0, scan_bytes = 0;
    GRID_TRY(hipcub::DeviceReduce::Min(nullptr, reduce_bytes, ulo, d_extent, n));
    GRID_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, counts, offsets, n));
    char *d_temp = nullptr;
    const size_t temp_bytes = std::max(reduce_bytes, scan_bytes);
    if (!buf.get(&d_temp, temp_bytes)) {
        why = "out of device memory";
        return false;
    }
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const double margin = kMarginRelDevice * scale;

    struct PerSample {
        int *start = nullptr;
        int2 *entries = nullptr;
        long long total = 0;
        int cells = 0;
    };
    std::vector<PerSample> done((size_t)num_samples);
    long long all_entries = 0, all_cells = 0;
    for (int k = 0; k < num_samples; ++k) {
        const float *dir = samples + 3 * (size_t)k;
        // the same basis as the host builder (shadow_grid.cpp)
        const double dx = dir[0], dy = dir[1], dz = dir[2];
        const double len2 = dx * dx + dy * dy + dz * dz;
        if (!std::isfinite(len2) || !(len2 > 1e-60)) {
            why = "degenerate sample direction";
            return false;
        }
        const double len = std::sqrt(len2);
        const double W[3] = {dx / len, dy / len, dz / len};
        int small = 0;
        if (std::fabs(W[1]) < std::fabs(W[small])) small = 1;
        if (std::fabs(W[2]) < std::fabs(W[small])) small = 2;
        double a[3] = {0, 0, 0};
        a[small] = 1.0;
        double U[3] = {a[1] * W[2] - a[2] * W[1], a[2] * W[0] - a[0] * W[2], a[0] * W[1] - a[1] * W[0]};
        const double ul = std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
        for (double &c : U) c /= ul;
        const double V[3] = {W[1] * U[2] - W[2] * U[1], W[2] * U[0] - W[0] * U[2], W[0] * U[1] - W[1] * U[0]};
        Basis bs;
        for (int c = 0; c < 3; ++c) bs.U[c] = (float)U[c], bs.V[c] = (float)V[c], bs.W[c] = (float)W[c];

        double tp = now();
        GRID_TRY(hipMemset(d_flag, 0, sizeof(int)));
        project_kernel<<<blocks, 256>>>((const TriRecord *)d_tris, n, bs, margin, ulo, uhi, vlo, vhi, zmax, d_flag);
        size_t tb = temp_bytes;
        GRID_TRY(hipcub::DeviceReduce::Min(d_temp, tb, ulo, d_extent + 0, n));
        tb = temp_bytes;
        GRID_TRY(hipcub::DeviceReduce::Max(d_temp, tb, uhi, d_extent + 1, n));
        tb = temp_bytes;
        GRID_TRY(hipcub::DeviceReduce::Min(d_temp, tb, vlo, d_extent + 2, n));
        tb = temp_bytes;
        GRID_TRY(hipcub::DeviceReduce::Max(d_temp, tb, vhi, d_extent + 3, n));
        float ext[4];
        int flag = 0;
        GRID_TRY(hipMemcpy(ext, d_extent, sizeof ext, hipMemcpyDeviceToHost));
        GRID_TRY(hipMemcpy(&flag, d_flag, sizeof flag, hipMemcpyDeviceToHost));
        if (flag) {
            why = "non-finite projection";
            return false;
        }
        ms_project += now() - tp, tp = now();
        const double gu0 = ext[0], gu1 = ext[1], gv0 = ext[2], gv1 = ext[3];
        const double extent = std::max(std::max(gu1 - gu0, gv1 - gv0), 1e-30);
        int res = (int)std::lround(std::sqrt(density * (double)n));
        res = std::min(std::max(res, 16), 4096);
        GridFrame g;
        CullFrame cf;
        cf.bs = bs, cf.margin = margin;
        long long total = 0;
        std::vector<unsigned> host_counts(N);
        for (;; res /= 2) { // halve the resolution until the lists fit
            g.inv_cell = (float)((double)res / (extent * (1.0 + 1e-6)));
            g.u0 = (float)gu0, g.v0 = (float)gv0, g.res = res;
            cf.cell = 1.0 / (double)g.inv_cell, cf.slack = cell_slack(margin, cf.cell);
            count_kernel<<<blocks, 256>>>((const TriRecord *)d_tris, n, g, cf, ulo, uhi, vlo, vhi, counts);
            GRID_TRY(hipMemcpy(host_counts.data(), counts, N * sizeof(unsigned), hipMemcpyDeviceToHost));
            total = 0;
            for (unsigned c : host_counts) total += c;
            if (total <= kMaxEntriesPerGridDevice) break;
            if (res <= 16) {
                why = "lists too long";
                return false;
            }
        }
        tb = temp_bytes;
        GRID_TRY(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, counts, offsets, n));

        GRID_TRY(hipDeviceSynchronize());
        ms_count += now() - tp, tp = now();
        const int cells = res * res;
        {
        Buffers tmp;
        unsigned long long *keys = nullptr, *keys_sorted = nullptr;
        int *vals = nullptr, *vals_sorted = nullptr;
        const size_t T = (size_t)std::max<long long>(total, 1);
        PerSample &ps = done[(size_t)k];
        if (!tmp.get(&keys, T) || !tmp.get(&keys_sorted, T) || !tmp.get(&vals, T) || !tmp.get(&vals_sorted, T) || !buf.get(&ps.start, (size_t)cells + 1) ||
            !buf.get(&ps.entries, T)) {
            why = "out of device memory";
            return false;
        }
        ms_alloc += now() - tp, tp = now();
        emit_kernel<<<blocks, 256>>>((const TriRecord *)d_tris, n, g, cf, ulo, uhi, vlo, vhi, zmax, counts, offsets, keys, vals);
        std::vector<int> wide;
        for (int i = 0; i < n; ++i)
            if (host_counts[(size_t)i] > kWideFace) wide.push_back(i);
        if (!wide.empty()) {
            int *d_wide = nullptr;
            if (!tmp.get(&d_wide, wide.size())) {
                why = "out of device memory";
                return false;
            }
            GRID_TRY(hipMemcpy(d_wide, wide.data(), wide.size() * sizeof(int), hipMemcpyHostToDevice));
            emit_wide_kernel<<<(unsigned)wide.size(), 256>>>(d_wide, g, ulo, uhi, vlo, vhi, zmax, offsets, keys, vals);
        }
        int cell_bits = 1;
        while ((1ll << cell_bits) < cells) ++cell_bits;
        size_t sort_bytes = 0;
        GRID_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys, keys_sorted, vals, vals_sorted, (int)total, 0, 32 + cell_bits));
        char *d_sort = nullptr;
        if (!tmp.get(&d_sort, sort_bytes)) {
            why = "out of device memory";
            return false;
        }
        if (total > 0) GRID_TRY(hipcub::DeviceRadixSort::SortPairs(d_sort, sort_bytes, keys, keys_sorted, vals, vals_sorted, (int)total, 0, 32 + cell_bits));
        GRID_TRY(hipDeviceSynchronize());
        ms_sort += now() - tp, tp = now();
        if (total > 0) entries_kernel<<<(unsigned)((total + 255) / 256), 256>>>(total, keys_sorted, vals_sorted, ps.entries);
        cell_start_kernel<<<(unsigned)((cells + 1 + 255) / 256), 256>>>(total, cells, keys_sorted, (int)all_entries, ps.start);
        GRID_TRY(hipGetLastError());
        GRID_TRY(hipDeviceSynchronize()); // `tmp` is freed at the end of this block
        ms_finish += now() - tp, tp = now();
        ps.total = total, ps.cells = cells;
        }
        ms_free += now() - tp;

        ShadowGridHeader h;
        std::memset(&h, 0, sizeof h);
        std::memcpy(h.U, bs.U, 12), std::memcpy(h.V, bs.V, 12), std::memcpy(h.W, bs.W, 12);
        h.u0 = g.u0, h.v0 = g.v0, h.inv_cell = g.inv_cell, h.res = res, h.first_cell = (int32_t)all_cells;
        out.headers.push_back(h);
        all_entries += total;
        all_cells += cells + 1;
        if (all_cells >= (1ll << 31) || all_entries >= (1ll << 31)) {
            why = "grids too large";
            return false;
        }
    }
    // one cell-start array and one entry array for all samples
    int *d_cells = nullptr;
    int2 *d_entries = nullptr;
    // (padding behind the last list: shadow_grid_kernel fetches a batch of entries whatever a list's length)
    if (!buf.get(&d_cells, (size_t)all_cells) || !buf.get(&d_entries, (size_t)std::max<long long>(all_entries, 1) + kShadowGridPadding)) {
        why = "out of device memory";
        return false;
    }
    GRID_TRY(hipMemset(d_entries + std::max<long long>(all_entries, 1) - 1, 0, (kShadowGridPadding + 1) * sizeof(int2))); // the padding names record 0
    long long cell_at = 0, entry_at = 0;
    for (const PerSample &ps : done) {
        GRID_TRY(hipMemcpy(d_cells + cell_at, ps.start, ((size_t)ps.cells + 1) * sizeof(int), hipMemcpyDeviceToDevice));
        if (ps.total > 0) GRID_TRY(hipMemcpy(d_entries + entry_at, ps.entries, (size_t)ps.total * sizeof(int2), hipMemcpyDeviceToDevice));
        cell_at += ps.cells + 1, entry_at += ps.total;
    }
    GRID_TRY(hipDeviceSynchronize());
    if (verbose)
        std::fprintf(stderr, "rtamd: device grids, %d samples: project+extent %.1f ms, count+scan %.1f, allocate %.1f, emit+sort %.1f, lists+cells %.1f, free %.1f\n",
                     num_samples, ms_project, ms_count, ms_alloc, ms_sort, ms_finish, ms_free);
    (void)ms_concat;
    buf.release(d_cells);
    buf.release(d_entries);
    out.d_cell_start = d_cells, out.d_entries = d_entries;
    out.num_cells = (uint64_t)all_cells, out.num_entries = (uint64_t)all_entries;
    out.usable = true;
    return true;
#undef GRID_TRY
}

} // namespace rtamd
