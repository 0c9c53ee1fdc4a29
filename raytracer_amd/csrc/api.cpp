// C ABI of the render pass (include/rtamd.h): scene upload, frame orchestration, statistics.
//
// A frame is a fixed sequence of launches with no host round trip:
//   memset(counters); for each SSAA pass, for depth 0..D: closest hits, compact, shadow rays, shade; [scale]
// (on the caller's stream; with reflections the deeper stages overlap on two side streams, see
// rtamd_render_device), which is the iterative form of Flyscene::traceRay's recursion (src/flyscene.cpp:581-667): the
// colour C = local + ks*LI*C_next is accumulated front to back with a running throughput.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtamd.h"
#include "bvh.hpp"
#include "lbvh.h"
#include "shadow_grid.hpp"
#include "device_types.h"
#include "kernels.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

using namespace rtamd;

namespace {

thread_local std::string g_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(RTAMD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct TimedLaunch {
    hipEvent_t start, stop;
    int kind; // 0 closest-hit walk, 1 shadow walk, 2 compaction, 3 shading, 4 other, 5 whole frame, 6 screen-bin pass
};

constexpr int kMaxFrameSlots = 32;
constexpr int kMaxHugeFaceList = 192; // kMaxHugeFaces of kernels.hip: entries of the list the compaction kernel tests against every pixel
// Page-locked report block of a frame slot: [4 * band + 0..3] what the screen-space pass of that band of rows found (a frame
// rendered whole is band 0; a frame has at most kMaxStages bands), [kStageFeedbackAt + 2 * stage + {0, 1}] hits and spawned
// rays of every stage, each + 1 (0: not reported yet).
constexpr int kVerdictInts = 4;
constexpr int kStageFeedbackAt = kVerdictInts * kMaxStages;
constexpr int kFeedbackInts = kStageFeedbackAt + 2 * kMaxStages;

// What one frame in flight owns.
struct FrameSlot {
    Workspace ws{};
    void *block = nullptr;
    size_t bins = 0;              // screen bins the workspace holds
    int depths = 0;               // recursion levels it was laid out for (1, 2, or 3 = any)
    size_t bytes = 0;             // size of `block`
    hipStream_t stream = nullptr; // the caller's stream the slot serves
    hipEvent_t done = nullptr;    // recorded behind the slot's most recent frame
    bool used = false;            // `done` has been recorded at least once
    uint64_t stamp = 0;           // scene clock at the most recent frame
    bool binned = false;          // the most recent frame ran the screen-bin passes
    bool rastered = false;        // ... or resolved its primary rays face by face
    // What the binning passes of earlier frames on this slot reported (pinned host memory the kernels write):
    // while it says that the lists were too long, the next frames of the same view walk the tree right away.
    volatile int32_t *feedback = nullptr;
    uint64_t feedback_key = 0;    // image size and partition the report belongs to
    uint64_t view_key = 0;        // ... and the camera and pass structure the stage counts belong to
    int frames_without_bins[kMaxStages] = {}; // per band of rows, like the verdicts
};

} // namespace

struct rtamd_scene {
    int device = -1;
    int num_cus = 256;
    void *d_nodes = nullptr, *d_tris = nullptr, *d_shade = nullptr, *d_materials = nullptr;
    void *d_tri_of_rank = nullptr, *d_cluster_box = nullptr, *d_planes = nullptr;
    void *d_light_pos = nullptr, *d_samples = nullptr, *d_spheres = nullptr, *d_texels = nullptr, *d_tex_info = nullptr;
    void *d_grid_headers = nullptr, *d_grid_cells = nullptr, *d_grid_entries = nullptr;
    DeviceScene ds{};
    // Frames in flight: every stream a caller renders on gets a workspace of its own (up to kMaxFrameSlots, then
    // the least recently used one is handed on), so frames enqueued on different streams may overlap on the GPU.
    FrameSlot slots[kMaxFrameSlots];
    int num_slots = 0;
    FrameSlot static_slot;   // frames of rtamd_render_device_noalloc: work buffers in the library's static device array
    char *scratch = nullptr; // that array's address on this scene's device
    FrameSlot *last_frame_slot = nullptr; // slot of the most recent frame (what rtamd_render_stats reports)
    uint64_t clock = 0;
    rtamd_scene_info info{};
    bool face_primary = true; // primary rays resolved face by face in screen space (rtamd_build_params::primary_rays AUTO)
    bool bin_primary = false; // screen binning of the primary rays (RTAMD_PRIMARY_RAYS_BINS)
    bool bin_always = false; // ... also where the tree would be cheaper (RTAMD_PRIMARY_RAYS_BINS)
    std::vector<rtamd_material> materials;
    std::vector<int32_t> face_material;
    // staging for the host-buffer entry point
    float *d_rgb = nullptr;
    int32_t *d_face = nullptr;
    size_t staging_pixels = 0;
    uint8_t *d_bytes = nullptr; // rtamd_render_ppm: the quantised frame, and behind it the int that says whether bytes suffice
    size_t bytes_pixels = 0;
    void *h_staging = nullptr; // page-locked host memory the planes are downloaded into when the caller's buffers are pageable
    size_t h_staging_bytes = 0;
    // bookkeeping of the most recent frame
    bool have_frame = false;
    rtamd_render_params last{};
    int last_stages = 0, last_depths = 0;
    uint64_t last_primary = 0;
    hipStream_t last_stream = nullptr;
    // A frame whose work buffers would exceed this is rendered in bands of rows, one after the other through buffers of one
    // band's size (rtamd_scene_set_workspace_limit; 0 = never)
    uint64_t workspace_limit = (uint64_t)2 << 30;
    size_t open_frame_event = 0; // timing event of a banded frame, opened by its first band
    // HIP-event pairs of every launch (kind 0..4) and frame (kind 5) since the last stats call
    std::vector<TimedLaunch> events;
    size_t events_used = 0;
};

namespace {

template <class T>
int upload(void **dst, const std::vector<T> &src, uint64_t *bytes) {
    const size_t n = src.size() * sizeof(T);
    HIP_TRY(hipMalloc(dst, n ? n : 64));
    if (n) HIP_TRY(hipMemcpy(*dst, src.data(), n, hipMemcpyHostToDevice));
    *bytes += n;
    return RTAMD_OK;
}

// Lays the work buffers of one frame in flight out behind `base` and returns the bytes they take; with base == nullptr
// nothing is written through `ws` but sizes, so the same code answers "how much" and "where".
// `slots` = ray slots of the local image = whole 8x8 tiles (a multiple of 64); bins == 0: no screen bins.
// `depths` = recursion levels the frames will have (1: no reflections): a frame without reflections needs no ray queues and
// one hit list, one with a single bounce one queue; three queues and both lists rotate from two bounces on.
size_t layout_workspace(char *base, size_t slots, size_t nshadow, size_t bins, size_t ntris, bool keys, int depths, Workspace &ws) {
    const size_t cap = (slots + 255) & ~(size_t)255; // every plane starts 1 KiB aligned
    const size_t head = kCounterHeadBytes;           // stage counters + work counters + shadow-ray counters
    static_assert(sizeof(StageCounters) * kMaxStages <= 2048, "counter block too small");
    static_assert(2048 + 3 * sizeof(WorkCounters) <= 4096 && 4096 + kShotSlots * kShotStride * 8 <= kCounterHeadBytes, "counter block too small");
    const size_t bin_capacity = bins ? std::min<size_t>(8 * ntris + (1u << 20), (size_t)1 << 30) : 0;
    uintptr_t p = (uintptr_t)base + head;
    auto take = [&](size_t bytes) {
        char *at = (char *)p;
        p += bytes;
        return at;
    };
    auto align = [&](uintptr_t to) { p = (p + to - 1) & ~(to - 1); };
    ws.stage = (StageCounters *)(uintptr_t)base;
    ws.work = (WorkCounters *)((uintptr_t)base + 2048);
    ws.shadow_shot = (unsigned long long *)((uintptr_t)base + 4096);
    ws.bin_state = nullptr, ws.bin_count = nullptr, ws.bin_start = nullptr, ws.bin_entries = nullptr, ws.big = nullptr;
    ws.pix_key = nullptr, ws.tile_flag = nullptr;
    ws.prefilled = 0;
    ws.bin_capacity = 0;
    if (bins || keys) { // right behind the counters: one sweep clears all of them at the start of a frame
        ws.bin_state = (BinState *)take(256);
        if (bins) ws.bin_count = (int32_t *)take((bins + 256) * 4);
        align(1024);
    }
    // tiles_x * 8 + tiles_y * 8 <= slots / 8 + 8 whatever the image's shape
    ws.dir_x = (float *)take((cap / 8 + 64) * 4);
    ws.dir_y = ws.dir_x; // placed per frame, behind the columns (enqueue_frame)
    align(1024);
    ws.slot_tri = (int32_t *)take(cap * 4);
    ws.slot_t = (float *)take(cap * 4);
    for (int k = 0; k < 2; ++k) { // by depth & 1
        HitList &h = ws.hits[k];
        if (k < depths) {
            h.slot = (int32_t *)take(cap * 4);
            h.tri = (int32_t *)take(cap * 4);
            h.child = (int32_t *)take(cap * 4);
            h.p = (float *)take(3 * cap * 4);
            ws.clear[k] = (uint8_t *)take(nshadow * cap);
            align(256);
        } else {
            h = ws.hits[0];
            ws.clear[k] = ws.clear[0];
        }
    }
    // queue[(d + 1) % 3] holds the rays spawned at depth d: one bounce only ever touches queue[1]
    for (int k = 0; k < 3; ++k) ws.queue[k] = nullptr;
    if (depths == 2) {
        ws.queue[1] = (float *)take(10 * cap * 4);
    } else if (depths > 2) {
        for (float *&q : ws.queue) q = (float *)take(10 * cap * 4);
    }
    if (keys) {
        align(256);
        ws.pix_key = (unsigned long long *)take(cap * 8);
        ws.tile_flag = (int32_t *)take(cap / 64 * 4); // right behind the keys: cleared together with them
    }
    if (bins || keys) {
        align(256);
        ws.big = (int4 *)take(((bins ? ntris : (size_t)kMaxHugeFaceList) + 1) * 16);
    }
    if (bins) {
        ws.bin_start = (int32_t *)take((bins + 256) * 4);
        ws.bin_entries = (int32_t *)take(bin_capacity * 4);
        ws.bin_capacity = (int32_t)bin_capacity;
    }
    ws.capacity = (int32_t)cap;
    return (size_t)(p - (uintptr_t)base) + 1024;
}

int ensure_workspace(rtamd_scene *s, FrameSlot &fs, size_t slots, size_t bins_needed, int depths) {
    if (fs.block && (size_t)fs.ws.capacity >= slots && fs.bins >= (s->bin_primary ? bins_needed : 0) && (fs.ws.pix_key != nullptr) == s->face_primary &&
        fs.depths >= std::min(depths, 3))
        return RTAMD_OK;
    if (fs.block) {
        HIP_TRY(hipFree(fs.block)); // waits for the device: no frame still reads the old block
        fs.block = nullptr;
    }
    const size_t nshadow = (size_t)s->ds.num_lights * (size_t)s->ds.spl;
    const size_t ntris = (size_t)s->info.num_device_faces;
    const size_t bins = s->bin_primary ? bins_needed : 0; // sized for the whole image this workspace may see
    int32_t *const feedback = fs.ws.bin_feedback, *const stage_feedback = fs.ws.stage_feedback;
    Workspace sizes{};
    const size_t total = layout_workspace(nullptr, slots, nshadow, bins, ntris, s->face_primary, depths, sizes);
    HIP_TRY(hipMalloc(&fs.block, total));
    layout_workspace((char *)fs.block, slots, nshadow, bins, ntris, s->face_primary, depths, fs.ws);
    if (fs.ws.pix_key) HIP_TRY(hipMemset(fs.ws.pix_key, 0, (size_t)fs.ws.capacity * sizeof(unsigned long long) + (size_t)fs.ws.capacity / 64 * 4));
    fs.ws.bin_feedback = feedback;
    fs.ws.stage_feedback = stage_feedback;
    fs.bins = bins;
    fs.depths = std::min(depths, 3);
    fs.bytes = total;
    return RTAMD_OK;
}

int check_params(const rtamd_scene *s, const rtamd_camera *cam, const rtamd_render_params *p) {
    if (!s || !cam || !p) return fail(RTAMD_ERR_INVALID, "null scene, camera or params");
    if (p->width <= 0 || p->height <= 0) return fail(RTAMD_ERR_INVALID, "image size %dx%d", p->width, p->height);
    if (p->ssaa_level < 1 || p->ssaa_level > 8) return fail(RTAMD_ERR_INVALID, "ssaa_level %d outside 1..8", p->ssaa_level);
    if (p->max_reflections < 0) return fail(RTAMD_ERR_INVALID, "max_reflections %d < 0", p->max_reflections);
    if (p->row_tile < 1 || p->num_parts < 1 || p->part < 0 || p->part >= p->num_parts)
        return fail(RTAMD_ERR_INVALID, "row partition (tile %d, parts %d, part %d)", p->row_tile, p->num_parts, p->part);
    if (!(cam->viewport[2] > 0) || !(cam->viewport[3] > 0)) return fail(RTAMD_ERR_INVALID, "empty viewport");
    return RTAMD_OK;
}

// The slot that serves `stream`.  When all slots are taken the least recently used one changes hands, after the
// host has waited for its last frame: no stream ever waits for another one here.
int acquire_slot(rtamd_scene *s, hipStream_t stream, FrameSlot **out) {
    int pick = -1;
    for (int i = 0; i < s->num_slots; ++i)
        if (s->slots[i].stream == stream) pick = i;
    if (pick < 0 && s->num_slots < kMaxFrameSlots) {
        pick = s->num_slots++;
        HIP_TRY(hipEventCreateWithFlags(&s->slots[pick].done, hipEventDisableTiming));
        void *host = nullptr, *mapped = nullptr;
        HIP_TRY(hipHostMalloc(&host, kFeedbackInts * sizeof(int32_t), hipHostMallocMapped));
        std::memset(host, 0, kFeedbackInts * sizeof(int32_t));
        HIP_TRY(hipHostGetDevicePointer(&mapped, host, 0));
        s->slots[pick].feedback = (volatile int32_t *)host;
        s->slots[pick].ws.bin_feedback = (int32_t *)mapped;
        s->slots[pick].ws.stage_feedback = (int32_t *)mapped + kStageFeedbackAt;
    }
    if (pick < 0) {
        pick = 0;
        for (int i = 1; i < s->num_slots; ++i)
            if (s->slots[i].stamp < s->slots[pick].stamp) pick = i;
        if (s->slots[pick].used) HIP_TRY(hipEventSynchronize(s->slots[pick].done));
    }
    FrameSlot &fs = s->slots[pick];
    fs.stream = stream;
    fs.stamp = ++s->clock;
    *out = &fs;
    return RTAMD_OK;
}

// ---- work buffers of rtamd_render_device_noalloc
// One array per GPU, part of the library's code object: every device that loads the library holds its own copy
// at its own address until the process ends.  Host code must never use the NAME as an address (on the host it is
// a stand-in symbol); the device address is looked up once per device, with that device current, and kept.
constexpr size_t kScratchBytes = (size_t)RTAMD_NOALLOC_WORKSPACE_BYTES;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
int scratch_pointer(int device, char **out) {
    static std::mutex lock;
    static char *per_device[kMaxDevices];
    if (device < 0 || device >= kMaxDevices) return fail(RTAMD_ERR_LIMIT, "device ordinal %d outside 0..%d", device, kMaxDevices - 1);
    std::lock_guard<std::mutex> guard(lock);
    if (!per_device[device]) {
        int current = -1;
        HIP_TRY(hipGetDevice(&current));
        if (current != device) return fail(RTAMD_ERR_INVALID, "device %d must be current to look up its work buffers (device %d is)", device, current);
        void *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(RTAMD_ERR_HIP, "hipGetSymbolAddress returned a null address for the static work buffers");
        per_device[device] = (char *)ptr;
    }
    *out = per_device[device];
    return RTAMD_OK;
}

bool any_reflective(const rtamd_scene *s, float threshold) {
    for (const rtamd_material &m : s->materials)
        if (m.ks[0] > threshold || m.ks[1] > threshold || m.ks[2] > threshold) return true;
    return false;
}

} // namespace

extern "C" {

const char *rtamd_last_error(void) { return g_error.c_str(); }

void rtamd_camera_default(rtamd_camera *cam, int32_t width, int32_t height) {
    if (!cam) return;
    std::memset(cam, 0, sizeof *cam);
    // Flycamera::reset: view = Translate(0, 0, -2)  ->  centre (0, 0, 2), inverse view = Translate(0, 0, 2)
    cam->origin[2] = 2.0f;
    cam->inv_view[0] = cam->inv_view[5] = cam->inv_view[10] = 1.0f;
    cam->inv_view[11] = 2.0f;
    cam->viewport[2] = (float)width;
    cam->viewport[3] = (float)height;
    cam->aspect_ratio = (float)width / (float)height;
    cam->fovy = 60.0f;
}

void rtamd_render_params_default(rtamd_render_params *p, int32_t width, int32_t height) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->width = width, p->height = height;
    p->ssaa_level = 1;
    p->max_reflections = 5;
    p->reflect_threshold = 0.25f;
    p->bounce_offset = 0.01f;
    p->t_epsilon = 0.00001f;
    p->background[0] = p->background[1] = p->background[2] = (float)0.4;
    p->row_tile = 16, p->num_parts = 1, p->part = 0;
}

int32_t rtamd_local_rows(const rtamd_render_params *p) {
    if (!p || p->row_tile < 1 || p->num_parts < 1 || p->part < 0 || p->part >= p->num_parts || p->height <= 0) return 0;
    const int ntiles = (p->height + p->row_tile - 1) / p->row_tile;
    if (p->part >= ntiles) return 0;
    const int tcount = (ntiles - p->part + p->num_parts - 1) / p->num_parts;
    int rows = tcount * p->row_tile;
    const int last_tile = p->part + (tcount - 1) * p->num_parts;
    if ((last_tile + 1) * p->row_tile > p->height) rows -= (last_tile + 1) * p->row_tile - p->height;
    return rows;
}

int32_t rtamd_global_row(const rtamd_render_params *p, int32_t local_row) {
    if (!p || p->row_tile < 1 || p->num_parts < 1) return -1;
    const int tile = local_row / p->row_tile;
    return (tile * p->num_parts + p->part) * p->row_tile + (local_row - tile * p->row_tile);
}

int rtamd_scene_create(const rtamd_scene_desc *d, const rtamd_build_params *build, rtamd_scene **out) {
    if (!d || !out) return fail(RTAMD_ERR_INVALID, "null scene description or output pointer");
    *out = nullptr;
    if (d->num_vertices < 0 || d->num_faces < 0 || d->num_materials < 0 || d->num_lights < 0 || d->num_spheres < 0 || d->samples_per_light < 0)
        return fail(RTAMD_ERR_INVALID, "negative count in scene description");
    if (d->num_faces > 0 && (!d->vertices || !d->normals || !d->face_vertex_ids || !d->face_normals || !d->face_origin_distance || !d->face_material_ids))
        return fail(RTAMD_ERR_INVALID, "null mesh array");
    if (d->num_faces >= (1 << 27)) return fail(RTAMD_ERR_LIMIT, "%d faces exceed the 2^27 limit", d->num_faces);
    if (d->num_unsampled_lights < 0 || d->num_unsampled_lights > d->num_lights)
        return fail(RTAMD_ERR_INVALID, "%d of %d lights without shadow samples", d->num_unsampled_lights, d->num_lights);
    const int32_t sampled_lights = d->num_lights - d->num_unsampled_lights; // PrecomputedData::lights.size()
    if ((d->num_materials > 0 && !d->materials) || (d->num_lights > 0 && !d->light_positions) ||
        (sampled_lights > 0 && d->samples_per_light > 0 && !d->light_samples) || (d->num_spheres > 0 && !d->spheres))
        return fail(RTAMD_ERR_INVALID, "null material, light or sphere array");
    if ((int64_t)sampled_lights * d->samples_per_light > kMaxShadowRays)
        return fail(RTAMD_ERR_LIMIT, "%d lights x %d samples exceed %d shadow rays per hit", sampled_lights, d->samples_per_light, kMaxShadowRays);
    for (int32_t f = 0; f < d->num_faces; ++f) {
        for (int k = 0; k < 3; ++k) {
            const int32_t id = d->face_vertex_ids[3 * (size_t)f + k];
            if (id < 0 || id >= d->num_vertices) return fail(RTAMD_ERR_INVALID, "face %d references vertex %d of %d", f, id, d->num_vertices);
        }
        const int32_t m = d->face_material_ids[f];
        if (m < -1 || m >= d->num_materials) return fail(RTAMD_ERR_INVALID, "face %d references material %d of %d", f, m, d->num_materials);
    }

    if (d->num_textures < 0) return fail(RTAMD_ERR_INVALID, "negative texture count");
    if (d->num_textures > 0) {
        if (!d->textures || !d->material_textures || !d->face_texcoords)
            return fail(RTAMD_ERR_INVALID, "textures given without texture coordinates or the material table");
        int64_t texel_total = 0;
        for (int32_t i = 0; i < d->num_textures; ++i) {
            const rtamd_texture &t = d->textures[i];
            if (t.width <= 0 || t.height <= 0 || !t.rgb) return fail(RTAMD_ERR_INVALID, "texture %d is empty", i);
            texel_total += (int64_t)t.width * t.height;
        }
        if (texel_total >= (1ll << 29)) return fail(RTAMD_ERR_LIMIT, "%lld texels exceed the 2^29 limit", (long long)texel_total);
        for (int32_t i = 0; i < d->num_materials; ++i)
            if (d->material_textures[i] < -1 || d->material_textures[i] >= d->num_textures)
                return fail(RTAMD_ERR_INVALID, "material %d references texture %d of %d", i, d->material_textures[i], d->num_textures);
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RTAMD_ERR_NO_DEVICE, "no HIP device available");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RTAMD_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", dev, prop.gcnArchName);

    rtamd_scene *s = new (std::nothrow) rtamd_scene();
    if (!s) return fail(RTAMD_ERR_INVALID, "out of host memory");
    s->device = dev;
    // (the array of rtamd_render_device_noalloc: looked up here, once per device, so that the entry point itself has
    // nothing left to set up; a scene that never uses that entry point does not depend on the lookup, and a failure is
    // reported by the entry point, which tries again)
    if (scratch_pointer(dev, &s->scratch) != RTAMD_OK) s->scratch = nullptr;
    s->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    // ---- host build
    const double t0 = now_ms();
    MeshView mesh{d->vertices, d->num_vertices, d->face_vertex_ids, d->face_normals, d->face_origin_distance, d->num_faces};
    std::vector<int32_t> rank;
    reference_face_rank(mesh, rank);
    FlatBvh bvh;
    const double th0 = now_ms();
    const bool verbose = std::getenv("RTAMD_VERBOSE_BUILD") != nullptr;
    int builder_used = RTAMD_BUILDER_HOST_SAH;
    if (build && build->builder == RTAMD_BUILDER_DEVICE_LBVH) {
        std::vector<TriRecord> records;
        gather_face_records(mesh, rank, records);
        std::string why;
        if (build_lbvh_device(records, build->max_leaf_faces, bvh, why)) builder_used = RTAMD_BUILDER_DEVICE_LBVH;
    } else if (build && build->builder != RTAMD_BUILDER_HOST_SAH) {
        delete s;
        return fail(RTAMD_ERR_INVALID, "unknown builder %d", build->builder);
    }
    if (builder_used == RTAMD_BUILDER_HOST_SAH)
        build_flat_bvh(mesh, rank, build ? build->max_leaf_faces : 0, build ? build->sah_bins : 0, bvh);
    const double th1 = now_ms();
    // The walk keeps at most stack_need entries; narrower nodes need fewer (a binary tree needs its depth).
    for (int width = 3; bvh.stack_need > kStackEntries && width >= 2 && !bvh.nodes.empty(); --width) collapse_bvh(bvh, width);
    if (bvh.stack_need > kStackEntries) {
        delete s;
        return fail(RTAMD_ERR_LIMIT, "BVH needs %d traversal stack entries, the kernels hold %d", bvh.stack_need, kStackEntries);
    }
    const bool textured = d->num_textures > 0 && d->face_texcoords && d->material_textures && d->textures;
    // The grids follow the hierarchy: built on the device when that was (after the records are uploaded, below),
    // on the host otherwise, or when the device build does not apply.
    ShadowGrids grids;
    const bool want_grids = !(build && build->shadow_rays == RTAMD_SHADOW_RAYS_TREE);
    bool grids_on_device = want_grids && builder_used == RTAMD_BUILDER_DEVICE_LBVH;
    if (want_grids && !grids_on_device) build_shadow_grids(bvh, d->light_samples, sampled_lights * d->samples_per_light, grids);
    const double tg = now_ms();
    // one record per TriRecord, in the same (leaf) order: hits that are neighbours in space read neighbouring records,
    // and the shading kernels need no detour over the face index
    const int32_t num_records = (int32_t)bvh.tris.size();
    if (bvh.tris.size() >= ((size_t)1 << 26)) { // the kernels address the 64-byte records by 32-bit byte offsets
        rtamd_scene_destroy(s);
        return fail(RTAMD_ERR_LIMIT, "%zu faces: at most %d are supported", bvh.tris.size(), (1 << 26) - 1);
    }
    std::vector<ShadeRecord> shade((size_t)num_records);
    auto fill_shade = [&](int32_t i0, int32_t i1) {
        for (int32_t i = i0; i < i1; ++i) {
            const int32_t f = bvh.tris[(size_t)i].face;
            ShadeRecord &r = shade[(size_t)i];
            std::memset(&r, 0, sizeof r);
            const int32_t *id = d->face_vertex_ids + 3 * (size_t)f;
            std::memcpy(r.v0, d->vertices + 3 * (size_t)id[0], 12);
            std::memcpy(r.v1, d->vertices + 3 * (size_t)id[1], 12);
            std::memcpy(r.v2, d->vertices + 3 * (size_t)id[2], 12);
            std::memcpy(r.n0, d->normals + 3 * (size_t)id[0], 12);
            std::memcpy(r.n1, d->normals + 3 * (size_t)id[1], 12);
            std::memcpy(r.n2, d->normals + 3 * (size_t)id[2], 12);
            std::memcpy(r.fn, d->face_normals + 3 * (size_t)f, 12);
            r.material = d->face_material_ids[f];
            if (textured) std::memcpy(r.tc, d->face_texcoords + 6 * (size_t)f, 24);
        }
    };
    if (num_records < (1 << 16)) {
        fill_shade(0, num_records);
    } else {
        const int workers = 8;
        std::vector<std::thread> pool;
        for (int w = 0; w < workers; ++w)
            pool.emplace_back(fill_shade, (int32_t)((int64_t)num_records * w / workers), (int32_t)((int64_t)num_records * (w + 1) / workers));
        for (std::thread &t : pool) t.join();
    }
    const double t1 = now_ms();
    if (verbose)
        std::fprintf(stderr, "rtamd: scene build, %d faces: face ranks %.1f ms, hierarchy %.1f ms, shadow grids %.1f ms, shading records %.1f ms\n",
                     d->num_faces, th0 - t0, th1 - th0, tg - th1, t1 - tg);

    // ---- upload
    uint64_t bytes = 0;
    int rc = RTAMD_OK;
    std::vector<float> mats((size_t)d->num_materials * kMaterialFloats);
    for (int32_t i = 0; i < d->num_materials; ++i) {
        float *m = &mats[(size_t)kMaterialFloats * (size_t)i];
        std::memcpy(m, d->materials[i].ka, 12);
        std::memcpy(m + 3, d->materials[i].kd, 12);
        std::memcpy(m + 6, d->materials[i].ks, 12);
        m[9] = d->materials[i].shininess;
        const int32_t tex = textured ? d->material_textures[i] : -1;
        std::memcpy(m + 10, &tex, 4);
    }
    std::vector<float> texels;
    std::vector<int32_t> tex_info;
    if (textured)
        for (int32_t i = 0; i < d->num_textures; ++i) {
            const rtamd_texture &t = d->textures[i];
            const int32_t first = (int32_t)(texels.size() / 3);
            tex_info.insert(tex_info.end(), {first, t.width, t.height, 0});
            texels.insert(texels.end(), t.rgb, t.rgb + (size_t)t.width * t.height * 3);
        }
    std::vector<float> lpos(d->light_positions, d->light_positions + 3 * (size_t)d->num_lights);
    std::vector<float> samples;
    if (sampled_lights > 0 && d->samples_per_light > 0) samples.assign(d->light_samples, d->light_samples + 3 * (size_t)sampled_lights * d->samples_per_light);
    std::vector<rtamd_sphere> spheres(d->spheres, d->spheres + d->num_spheres);
    if (rc == RTAMD_OK) rc = upload(&s->d_nodes, bvh.nodes4, &bytes);
    if (rc == RTAMD_OK) rc = upload(&s->d_tris, bvh.tris, &bytes);
    if (rc == RTAMD_OK) rc = upload(&s->d_shade, shade, &bytes);
    // The plane of every record once more, 16 bytes apiece: a shadow ray tests the planes of all its candidates and the
    // corners of few, and four planes share the 64-byte line that holds one whole record.
    std::vector<float> planes((size_t)num_records * 4);
    for (int32_t i = 0; i < num_records; ++i) std::memcpy(&planes[(size_t)i * 4], bvh.tris[(size_t)i].n, 16);
    if (rc == RTAMD_OK) rc = upload(&s->d_planes, planes, &bytes);
    std::vector<int32_t> tri_of_rank((size_t)d->num_faces, -1); // face raster: the pixel keeps a rank, the shading wants the record
    for (int32_t i = 0; i < num_records; ++i) {
        const int32_t r = bvh.tris[(size_t)i].rank;
        if (r >= 0 && r < d->num_faces) tri_of_rank[(size_t)r] = i;
    }
    if (rc == RTAMD_OK) rc = upload(&s->d_tri_of_rank, tri_of_rank, &bytes);
    // Boxes of every 64 consecutive records (neighbours in the hierarchy, so neighbours in space): a rank that renders only
    // some rows of the image drops whole groups of faces that lie in other ranks' rows before it touches their records.
    std::vector<float> cluster_box((size_t)((num_records + 63) / 64) * 8, 0.0f);
    for (int32_t c = 0; c * 64 < num_records; ++c) {
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int32_t i = c * 64; i < std::min(num_records, (c + 1) * 64); ++i) {
            const TriRecord &t = bvh.tris[(size_t)i];
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::min(lo[k], std::min(t.v0[k], std::min(t.v1[k], t.v2[k])));
                hi[k] = std::max(hi[k], std::max(t.v0[k], std::max(t.v1[k], t.v2[k])));
            }
        }
        for (int k = 0; k < 3; ++k) cluster_box[(size_t)c * 8 + k] = lo[k], cluster_box[(size_t)c * 8 + 4 + k] = hi[k];
    }
    if (rc == RTAMD_OK) rc = upload(&s->d_cluster_box, cluster_box, &bytes);
    if (rc == RTAMD_OK) rc = upload(&s->d_materials, mats, &bytes);
    if (rc == RTAMD_OK) rc = upload(&s->d_light_pos, lpos, &bytes);
    if (rc == RTAMD_OK) rc = upload(&s->d_samples, samples, &bytes);
    if (rc == RTAMD_OK) rc = upload(&s->d_spheres, spheres, &bytes);
    if (rc == RTAMD_OK) rc = upload(&s->d_texels, texels, &bytes);
    if (rc == RTAMD_OK) rc = upload(&s->d_tex_info, tex_info, &bytes);
    DeviceShadowGrids dgrids;
    double grid_ms = 0.0;
    if (rc == RTAMD_OK && grids_on_device) {
        const double tg0 = now_ms();
        std::string why;
        if (build_shadow_grids_device(bvh.tris, s->d_tris, d->light_samples, sampled_lights * d->samples_per_light, shadow_grid_density(bvh.tris.size()), dgrids, why)) {
            s->d_grid_cells = dgrids.d_cell_start, s->d_grid_entries = dgrids.d_entries;
            bytes += dgrids.num_cells * 4 + dgrids.num_entries * sizeof(ShadowGridEntry);
            rc = upload(&s->d_grid_headers, dgrids.headers, &bytes);
        } else {
            grids_on_device = false;
            build_shadow_grids(bvh, d->light_samples, sampled_lights * d->samples_per_light, grids);
        }
        grid_ms = now_ms() - tg0;
        if (verbose) std::fprintf(stderr, "rtamd: shadow grids on the %s: %.1f ms%s%s\n", grids_on_device ? "device" : "host", grid_ms, why.empty() ? "" : " -- ", why.c_str());
    }
    if (!grids_on_device) {
        if (rc == RTAMD_OK) rc = upload(&s->d_grid_headers, grids.headers, &bytes);
        if (rc == RTAMD_OK) rc = upload(&s->d_grid_cells, grids.cell_start, &bytes);
        // (padding: the kernel fetches a batch of a few entries whatever the length of the list)
        grids.entries.resize(grids.entries.size() + kShadowGridPadding, ShadowGridEntry{});
        if (rc == RTAMD_OK) rc = upload(&s->d_grid_entries, grids.entries, &bytes);
    }
    if (rc != RTAMD_OK) {
        rtamd_scene_destroy(s);
        return rc;
    }
    const double t2 = now_ms();

    s->ds.nodes = (const float4 *)s->d_nodes;
    s->ds.tris = (const float4 *)s->d_tris;
    s->ds.planes = (const float4 *)s->d_planes;
    s->ds.num_records = num_records;
    float largest = 0.0f; // largest coordinate of any face
    for (const TriRecord &t : bvh.tris)
        for (int k = 0; k < 3; ++k) largest = std::max(largest, std::max(std::fabs(t.v0[k]), std::max(std::fabs(t.v1[k]), std::fabs(t.v2[k]))));
    static const float slack_rel = getenv("RTAMD_BOX_SLACK") ? (float)atof(getenv("RTAMD_BOX_SLACK")) : 9.5367431640625e-7f; // experiment hook (2^-20)
    s->ds.box_slack = std::isfinite(largest) ? largest * slack_rel : 0.0f;
    s->ds.box_slack_rel = slack_rel;
    s->ds.largest_coord = std::isfinite(largest) ? largest : 0.0f;
    s->ds.shade = (const float4 *)s->d_shade;
    s->ds.materials = (const float *)s->d_materials;
    s->ds.light_pos = (const float *)s->d_light_pos;
    s->ds.samples = (const float *)s->d_samples;
    s->ds.spheres = (const float4 *)s->d_spheres;
    s->ds.tri_of_rank = (const int32_t *)s->d_tri_of_rank;
    s->ds.cluster_box = (const float4 *)s->d_cluster_box;
    s->ds.texels = (const float *)s->d_texels;
    s->ds.tex_info = (const int32_t *)s->d_tex_info;
    s->ds.num_textures = textured ? d->num_textures : 0;
    s->ds.grid_headers = (const float4 *)s->d_grid_headers;
    s->ds.grid_cells = (const int32_t *)s->d_grid_cells;
    s->ds.grid_entries = (const int2 *)s->d_grid_entries;
    const bool grids_usable = grids_on_device ? dgrids.usable : grids.usable;
    s->ds.use_grids = grids_usable ? 1 : 0;
    for (int ci = 0; ci < 4; ++ci) {
        const int32_t c = bvh.nodes4[0].child[ci];
        for (int gk = 0; gk < 4; ++gk) // a leaf directly under the root goes to one lane
            s->ds.start16[4 * ci + gk] = c >= 0 ? bvh.nodes4[(size_t)c].child[gk] : (gk == 0 ? c : 0x7fffffff);
    }
    static const bool faces_off = getenv("RTAMD_PRIMARY_FACES") && atoi(getenv("RTAMD_PRIMARY_FACES")) == 0; // experiment hook: AUTO bins as it used to
    const int primary_mode = build ? build->primary_rays : RTAMD_PRIMARY_RAYS_AUTO;
    s->face_primary = primary_mode == RTAMD_PRIMARY_RAYS_AUTO && !faces_off;
    s->bin_primary = primary_mode == RTAMD_PRIMARY_RAYS_BINS || (primary_mode == RTAMD_PRIMARY_RAYS_AUTO && faces_off);
    s->bin_always = primary_mode == RTAMD_PRIMARY_RAYS_BINS;
    s->info.shadow_grid_entries = !grids_usable ? 0 : grids_on_device ? dgrids.num_entries : (uint64_t)grids.entries.size() - kShadowGridPadding;
    s->ds.num_lights = sampled_lights;
    s->ds.num_shade_lights = d->num_lights;
    s->ds.spl = d->samples_per_light;
    s->ds.num_spheres = d->num_spheres;
    s->ds.num_materials = d->num_materials;
    s->materials.assign(d->materials, d->materials + d->num_materials);

    s->info.num_faces = d->num_faces;
    s->info.num_device_faces = bvh.num_device_faces;
    s->info.num_nodes = (int32_t)bvh.nodes4.size();
    s->info.num_leaves = bvh.num_leaves;
    s->info.depth = bvh.depth4;
    s->info.device = dev;
    s->info.device_bytes = bytes;
    s->info.build_ms = (float)(t1 - t0 + grid_ms);
    s->info.builder = builder_used;
    s->info.hierarchy_ms = (float)(th1 - th0);
    s->info.upload_ms = (float)(t2 - t1);
    *out = s;
    return RTAMD_OK;
}

int rtamd_reference_face_rank(const rtamd_scene_desc *d, int32_t *rank_out) {
    if (!d || !rank_out) return fail(RTAMD_ERR_INVALID, "null scene description or output");
    if (d->num_faces < 0 || d->num_vertices < 0) return fail(RTAMD_ERR_INVALID, "negative count in scene description");
    if (d->num_faces > 0 && (!d->vertices || !d->face_vertex_ids)) return fail(RTAMD_ERR_INVALID, "null mesh array");
    for (int64_t i = 0; i < 3 * (int64_t)d->num_faces; ++i)
        if (d->face_vertex_ids[i] < 0 || d->face_vertex_ids[i] >= d->num_vertices)
            return fail(RTAMD_ERR_INVALID, "face %lld references vertex %d of %d", (long long)(i / 3), d->face_vertex_ids[i], d->num_vertices);
    MeshView mesh{d->vertices, d->num_vertices, d->face_vertex_ids, d->face_normals, d->face_origin_distance, d->num_faces};
    std::vector<int32_t> rank;
    reference_face_rank(mesh, rank);
    std::memcpy(rank_out, rank.data(), rank.size() * sizeof(int32_t));
    return RTAMD_OK;
}

int rtamd_scene_destroy(rtamd_scene *s) {
    if (!s) return RTAMD_OK;
    void *bufs[] = {s->d_tri_of_rank, s->d_cluster_box, s->d_planes, s->d_nodes, s->d_tris,  s->d_shade,  s->d_materials, s->d_light_pos, s->d_samples,
                    s->d_spheres, s->d_texels, s->d_tex_info, s->d_grid_headers, s->d_grid_cells, s->d_grid_entries,
                    s->d_rgb, s->d_face, s->d_bytes};
    (void)hipDeviceSynchronize(); // frames may still be in flight on the caller's streams
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (int i = 0; i < s->num_slots; ++i) {
        FrameSlot &fs = s->slots[i];
        if (fs.block) (void)hipFree(fs.block);
        if (fs.done) (void)hipEventDestroy(fs.done);
        if (fs.feedback) (void)hipHostFree((void *)fs.feedback);
    }
    if (s->h_staging) (void)hipHostFree(s->h_staging);
    for (TimedLaunch &t : s->events) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    delete s;
    return RTAMD_OK;
}

int rtamd_scene_set_lights(rtamd_scene *s, const float *positions, const float *colors, int32_t num_lights) {
    (void)colors; // carried by the boundary, never applied (src/flyscene.cpp:552 builds minSum and drops it)
    if (!s) return fail(RTAMD_ERR_INVALID, "null scene");
    if (num_lights < s->ds.num_lights) return fail(RTAMD_ERR_INVALID, "%d lights given, %d of the scene's lights have shadow samples", num_lights, s->ds.num_lights);
    if (num_lights > 0 && !positions) return fail(RTAMD_ERR_INVALID, "null light positions");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != s->device) return fail(RTAMD_ERR_INVALID, "scene lives on device %d but device %d is current", s->device, dev);
    void *fresh = nullptr;
    const size_t bytes = 3 * sizeof(float) * (size_t)num_lights;
    HIP_TRY(hipMalloc(&fresh, bytes ? bytes : 64));
    if (bytes && hipMemcpy(fresh, positions, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(fresh);
        return fail(RTAMD_ERR_HIP, "upload of %d light positions failed", num_lights);
    }
    HIP_TRY(hipDeviceSynchronize()); // frames in flight still read the old list
    if (s->d_light_pos) (void)hipFree(s->d_light_pos);
    s->d_light_pos = fresh;
    s->ds.light_pos = (const float *)fresh;
    s->ds.num_shade_lights = num_lights;
    return RTAMD_OK;
}

int rtamd_scene_get_info(const rtamd_scene *s, rtamd_scene_info *info) {
    if (!s || !info) return fail(RTAMD_ERR_INVALID, "null scene or info");
    *info = s->info;
    return RTAMD_OK;
}

} // extern "C"

namespace {

// What a frame of the given parameters needs; filled by frame_shape.
struct FrameShape {
    int local_rows = 0, S = 1, depths = 1, stages = 1, tiles_x = 0, tiles_y = 0, bins_x = 0, bins_y = 0;
    size_t pixels = 0, slots = 0, nshadow = 0;
};

int frame_shape(const rtamd_scene *s, const rtamd_render_params *p, FrameShape &f) {
    f.local_rows = rtamd_local_rows(p);
    f.pixels = (size_t)f.local_rows * (size_t)p->width;
    f.S = p->ssaa_level;
    f.depths = (any_reflective(s, p->reflect_threshold) ? p->max_reflections : 0) + 1;
    f.stages = f.S * f.S * f.depths;
    if (f.stages > kMaxStages) return fail(RTAMD_ERR_LIMIT, "ssaa^2 * (max_reflections + 1) = %d exceeds %d", f.stages, kMaxStages);
    f.tiles_x = (p->width + 7) / 8, f.tiles_y = (f.local_rows + 7) / 8;
    f.slots = (size_t)f.tiles_x * (size_t)f.tiles_y * 64;
    f.nshadow = (size_t)s->ds.num_lights * (size_t)s->ds.spl;
    if (f.slots * (f.nshadow ? f.nshadow : 1) > (size_t)0x7f000000)
        return fail(RTAMD_ERR_LIMIT, "%zu ray slots x %zu shadow rays per hit exceed the 2^31 work-item limit of one launch", f.slots, f.nshadow);
    f.bins_x = (p->width + kBinPixels - 1) / kBinPixels, f.bins_y = (p->height + kBinPixels - 1) / kBinPixels;
    return RTAMD_OK;
}

// Rows of one band when an image of `height` rows is rendered in `bands` bands: whole 8-row tiles, the last band
// takes what is left.
int band_height(int height, int bands) { return ((height + bands - 1) / bands + 7) & ~7; }

// In how many bands of rows a frame is rendered: one, unless the frame is an unpartitioned image whose work buffers would
// exceed the scene's limit; then as few as bring one band's buffers under it, while the stage counters last (every band
// takes a frame's worth of them) and every band still has rows.
// (`limit` bytes for a workspace with `bins` screen bins and, if `keys`, the per-pixel keys of the face pass; 0 = no bound)
int plan_bands(const rtamd_scene *s, const rtamd_render_params *p, const FrameShape &shape, size_t limit, size_t bins, bool keys) {
    if (limit == 0 || p->num_parts != 1) return 1;
    Workspace sizes{};
    const size_t ntris = (size_t)s->info.num_device_faces;
    const size_t whole = layout_workspace(nullptr, shape.slots, shape.nshadow, bins, ntris, keys, shape.depths, sizes);
    if (whole <= limit) return 1;
    int bands = 1;
    const int most = std::min(kMaxStages / std::max(shape.stages, 1), (p->height + 7) / 8);
    while (bands < most) {
        ++bands;
        const int rows = band_height(p->height, bands);
        if ((size_t)(bands - 1) * (size_t)rows >= (size_t)p->height) { // the last band would be empty: not a usable count
            continue;
        }
        const size_t slots = (size_t)shape.tiles_x * (size_t)((rows + 7) / 8) * 64;
        if (layout_workspace(nullptr, slots, shape.nshadow, bins, ntris, keys, shape.depths, sizes) <= limit) break;
    }
    while (bands > 1 && (size_t)(bands - 1) * (size_t)band_height(p->height, bands) >= (size_t)p->height) --bands;
    return bands;
}

// Enqueues every launch of one frame on `stream`, using the work buffers of `fs`.  `plain`: the caller may not
// allocate or create anything -- no timing events, no completion event.
// A frame rendered in bands (rtamd_render_device) comes here once per band: `p` describes the band as part `band` of
// `bands` (row tiles of one band's height), `whole` the frame the caller asked for; the outputs point at the band's first
// row.  Band b counts its stages from b * stages on, so that the counters of all bands add up to the frame's.
int enqueue_frame(rtamd_scene *s, FrameSlot &fs, const FrameShape &shape, const rtamd_camera *cam, const rtamd_render_params *p, float *rgb,
                  int32_t *face_id, float *t_hit, hipStream_t stream, bool plain, int band = 0, int bands = 1,
                  const rtamd_render_params *whole = nullptr) {
    int rc = RTAMD_OK;
    if (!whole) whole = p;
    const int stage_base = band * shape.stages;
    const int local_rows = shape.local_rows, S = shape.S, depths = shape.depths, stages = shape.stages;
    const int tiles_x = shape.tiles_x, tiles_y = shape.tiles_y, bins_x = shape.bins_x, bins_y = shape.bins_y;
    const size_t pixels = shape.pixels, nshadow = shape.nshadow;

    FrameConst fc;
    std::memset(&fc, 0, sizeof fc);
    std::memcpy(fc.origin, cam->origin, sizeof fc.origin);
    std::memcpy(fc.inv_view, cam->inv_view, sizeof fc.inv_view);
    std::memcpy(fc.vp, cam->viewport, sizeof fc.vp);
    // Camera::getPerspectiveScale and its reciprocal, camera.hpp:168,271-274 (double intermediates)
    const float persp = (float)((double)1.0f / std::tan((double)(cam->fovy / 2.0f) * (M_PI / (double)180.0f)));
    fc.scale = (float)(1.0 / (double)persp);
    fc.ax = cam->aspect_ratio * fc.scale;
    fc.width = p->width, fc.height = p->height, fc.local_rows = local_rows;
    fc.tiles_x = tiles_x, fc.tiles_y = tiles_y;
    fc.row_tile = p->row_tile, fc.num_parts = p->num_parts, fc.part = p->part;
    fc.tiles_x_magic = tiles_x == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint64_t)tiles_x);
    fc.num_parts_magic = p->num_parts == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint64_t)p->num_parts);
    fc.row_tile_magic = p->row_tile == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (uint64_t)p->row_tile);
    fc.t_eps = p->t_epsilon, fc.refl_thresh = p->reflect_threshold, fc.bounce_off = p->bounce_offset;
    std::memcpy(fc.bg, p->background, sizeof fc.bg);
    std::memcpy(fc.bg_deep, p->background_deep, sizeof fc.bg_deep);

    // inverse of camera_direction for the screen binning: r = s * L * (n + e), see project_point in kernels.hip
    // binning pays unless the faces are many against the pixels this rank renders: its cost grows with the faces,
    // the tree walk's with the pixels.  Measured on a 1.1 M-face field: still ahead at one face per pixel (one
    // rank of eight at 4K), behind at 7.5 faces per pixel (a 256x144 image of 278 K faces).
    static const double faces_per_pixel_max = getenv("RTAMD_BIN_FACES_PER_PIXEL") ? atof(getenv("RTAMD_BIN_FACES_PER_PIXEL")) : 2.0;
    bool bins_kept_before = false;
    bool bins_ok = s->bin_primary && fs.ws.bin_entries != nullptr && bins_x < 32768 && bins_y < 32768 &&
                   (s->bin_always || (double)s->info.num_device_faces <= faces_per_pixel_max * (double)pixels);
    // Face by face, the pass costs a projection per face and a test per pixel of each face's box; the tree walk costs some
    // twenty-five node visits per pixel.  Past eight faces per rendered pixel the projections alone outweigh the walk.
    static const double faces_per_pixel_raster = getenv("RTAMD_RASTER_FACES_PER_PIXEL") ? atof(getenv("RTAMD_RASTER_FACES_PER_PIXEL")) : 8.0;
    bool faces_ok = s->face_primary && fs.ws.pix_key != nullptr && p->width < 65536 && p->height < 65536 && p->t_epsilon > 0.0f && // (distances are ordered by their bits: -0.0f must not pass)
                    (double)s->info.num_device_faces <= faces_per_pixel_raster * (double)pixels;
    // What earlier frames on this slot reported (page-locked memory their kernels wrote; a frame still in flight may be
    // half-way through updating it, which is harmless: the figures only pick launch sizes and the cheaper of two
    // equivalent paths).  Stage counts carry over between frames of one image size, partition and pass structure; the
    // verdict on the screen bins only between frames of one view, camera included.
    auto mix = [](uint64_t h, const void *data, size_t bytes) {
        const unsigned char *b = (const unsigned char *)data;
        for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull; // FNV-1a
        return h;
    };
    const int32_t shape_words[9] = {p->width, p->height, whole->part, whole->num_parts, whole->row_tile, S, depths, (int32_t)nshadow, bands};
    const uint64_t size_key = mix(14695981039346656037ull, shape_words, sizeof shape_words) | 1ull;
    const uint64_t view_key = mix(size_key, cam, sizeof *cam) | 1ull;
    std::vector<int> expect_hits((size_t)kMaxStages, -1), expect_next((size_t)kMaxStages, -1); // by stage of the frame
    if (fs.feedback) {
        if (fs.feedback_key != size_key) {
            fs.feedback_key = size_key;
            for (int i = kStageFeedbackAt; i < kFeedbackInts; ++i) fs.feedback[i] = 0;
        }
        static const bool sized = !getenv("RTAMD_WORST_CASE_LAUNCHES"); // experiment hook
        for (int i = stage_base; sized && i < stage_base + stages; ++i) {
            expect_hits[(size_t)i] = fs.feedback[kStageFeedbackAt + 2 * i] - 1;
            expect_next[(size_t)i] = fs.feedback[kStageFeedbackAt + 2 * i + 1] - 1;
        }
    }
    // Every band of rows keeps a verdict of its own: the faces too large for the face pass are counted per band, so one
    // band may give way to the tree walk while the next one does not.
    volatile int32_t *const verdict = fs.feedback ? fs.feedback + kVerdictInts * band : nullptr;
    if (fs.feedback && band == 0 && fs.view_key != view_key) { // another view: what earlier passes reported says nothing about it
        fs.view_key = view_key;
        for (int i = 0; i < kStageFeedbackAt; ++i) fs.feedback[i] = 0;
        for (int &n : fs.frames_without_bins) n = 0;
    }
    if ((bins_ok || faces_ok) && verdict && !s->bin_always) {
        const bool turned_down = verdict[3] != 0 && verdict[2] != 0; // [2]: the pass's own verdict (bins_accepted)
        bins_kept_before = verdict[3] != 0 && !turned_down;
        if (turned_down && fs.frames_without_bins[band] < 63) { // try the bins again every 64th frame: the camera may have moved
            bins_ok = faces_ok = false;
            ++fs.frames_without_bins[band];
        } else {
            fs.frames_without_bins[band] = 0;
        }
    }
    // Does the view leave most pixels untouched?  Then the output planes are filled with the miss values in one sweep and
    // the compaction looks at the touched tiles only (prep_kernel); where most pixels are hit, or faces too large for the
    // face pass must be tested pixel by pixel anyway, that sweep is wasted (1.1 M-face field at 4K: 10 % of the frame).
    // Without a report from an earlier frame of the view the sparse form is assumed.
    bool sparse_view = true;
    if (verdict && fs.view_key == view_key && verdict[3] != 0 && verdict[1] > 0) sparse_view = false;
    if (expect_hits[(size_t)stage_base] >= 0 && 2 * (size_t)expect_hits[(size_t)stage_base] >= pixels) sparse_view = false;
    if (bins_ok || faces_ok) {
        const float *m = cam->inv_view;
        const double a = m[0], b = m[1], c = m[2], d0 = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
        const double c00 = e * i - f * h, c01 = f * g - d0 * i, c02 = d0 * h - e * g;
        const double det = a * c00 + b * c01 + c * c02;
        if (!(std::fabs(det) > 1e-30) || !std::isfinite(det)) {
            bins_ok = faces_ok = false;
        } else {
            const double r = 1.0 / det;
            const double inv[9] = {c00 * r, (c * h - b * i) * r, (b * f - c * e) * r, c01 * r, (a * i - c * g) * r,
                                   (c * d0 - a * f) * r, c02 * r, (b * g - a * h) * r, (a * e - b * d0) * r};
            const double dt[3] = {(double)m[3] - cam->origin[0], (double)m[7] - cam->origin[1], (double)m[11] - cam->origin[2]};
            for (int k = 0; k < 9; ++k) fc.lin_inv[k] = (float)inv[k];
            for (int k = 0; k < 3; ++k) fc.cam_e[k] = (float)(inv[3 * k] * dt[0] + inv[3 * k + 1] * dt[1] + inv[3 * k + 2] * dt[2]);
            if (!(std::fabs(fc.cam_e[2] - 1.0f) > 1e-6f)) bins_ok = faces_ok = false;
            // the divisions of the projection, done once here: project_point multiplies
            fc.proj_s = (float)(1.0 / ((double)fc.cam_e[2] - 1.0));
            fc.px_off = (float)((double)fc.vp[0] + 0.5 * (double)fc.vp[2]), fc.px_scale = (float)(0.5 * (double)fc.vp[2] / (double)fc.ax);
            fc.py_off = (float)((double)fc.vp[1] + 0.5 * (double)fc.vp[3]), fc.py_scale = (float)(0.5 * (double)fc.vp[3] / (double)fc.scale);
            if (!std::isfinite(fc.proj_s) || !std::isfinite(fc.px_scale) || !std::isfinite(fc.py_scale)) bins_ok = faces_ok = false;
        }
        fc.bins_x = bins_x, fc.bins_y = bins_y;
        // four times the worst-case error of a projected on-screen corner (derivation: RT_BIN_MARGIN in kernels.hip), with the
        // scale of the projection taken from this camera: pixels per unit of n = (H / 2) / tan(fovy / 2)
        const double unit = 5.9604644775390625e-8; // 2^-24
        const double px_per_n = std::max(0.5 * (double)p->height / std::max(1e-6, (double)std::fabs(fc.scale)),
                                         0.5 * (double)p->width / std::max(1e-6, (double)std::fabs(fc.ax)));
        const double bound = px_per_n * 32.0 * unit + 6.0 * unit * (double)p->width;
        fc.raster_margin = std::max(raster_margin_floor(), (float)(4.0 * bound));
        // the exact test's own rounding: four units in the last place of the largest coordinate it computes with (eye or face),
        // times four, in pixels per unit of n (project_point's `spread` turns it into pixels at a corner's depth)
        const double largest = std::max((double)s->ds.largest_coord, std::max(std::fabs((double)cam->origin[0]), std::max(std::fabs((double)cam->origin[1]), std::fabs((double)cam->origin[2]))));
        static const double coord_rel = getenv("RTAMD_COORD_SLACK") ? atof(getenv("RTAMD_COORD_SLACK")) : 1.9073486328125e-6; // experiment hook (2^-19)
        fc.coord_slack = (float)(coord_rel * largest * std::max((double)std::fabs(fc.px_scale), (double)std::fabs(fc.py_scale)));
    }
    fs.ws.dir_y = fs.ws.dir_x + tiles_x * 8;
    Workspace ws_frame = fs.ws; // what the kernels of this frame see
    if (ws_frame.bin_feedback) ws_frame.bin_feedback += kVerdictInts * band;
    if (!bins_ok) ws_frame.bin_entries = nullptr;
    if (!faces_ok) ws_frame.pix_key = nullptr;
    fs.binned = bins_ok;
    fs.rastered = faces_ok;
    ws_frame.bin_max_list = s->bin_always ? 0x7fffffff : kMaxAverageBinList;
    ws_frame.num_bins = bins_x * bins_y;
    ws_frame.bins_expected = (bins_ok || faces_ok) && bins_kept_before;
    ws_frame.listed_faces_expected = s->ds.num_spheres > 0 || (verdict && fs.view_key == view_key && verdict[3] != 0 && verdict[1] > 0);
    static const int forced_waves = getenv("RTAMD_FULL_WAVES") ? atoi(getenv("RTAMD_FULL_WAVES")) : -1; // experiment hook
    ws_frame.full_waves = forced_waves >= 0 ? forced_waves : s->num_slots > 1; // several caller streams: frames overlap

    const bool count = (p->flags & RTAMD_FLAG_COUNT_WORK) != 0;
    const bool timed = !plain && (p->flags & RTAMD_FLAG_TIME_KERNELS) != 0;
    size_t open_slot = 0;
    auto begin = [&](int kind) -> int {
        if (!timed) return RTAMD_OK;
        if (s->events_used == s->events.size()) {
            TimedLaunch t;
            t.kind = kind;
            HIP_TRY(hipEventCreate(&t.start));
            HIP_TRY(hipEventCreate(&t.stop));
            s->events.push_back(t);
        }
        open_slot = s->events_used++;
        s->events[open_slot].kind = kind;
        HIP_TRY(hipEventRecord(s->events[open_slot].start, stream));
        return RTAMD_OK;
    };
    auto end = [&]() -> int {
        if (!timed) return RTAMD_OK;
        HIP_TRY(hipEventRecord(s->events[open_slot].stop, stream));
        return RTAMD_OK;
    };
    if (timed && band == 0) {
        if ((rc = begin(5)) != RTAMD_OK) return rc;
        s->open_frame_event = open_slot;
    }

    // Per SSAA pass the work of depth d is walk(d) -> compact(d) (which also spawns the rays of depth d+1) ->
    // shadow walk(d) -> shade(d).  Only shade(d) needs shade(d-1) (it hands down the throughput); walk(d+1) and
    // shadow walk(d) both need just compact(d), and compact(d+1) and shade(d) do not touch each other's data.
    // Buffers rotate (hit lists and shadow results by d & 1, queues by d % 3): compact(d + 2) is the first kernel to
    // overwrite what shade(d) reads.  Everything is enqueued on the caller's stream: independent pieces share a
    // launch (below) instead of running on streams of the library's own, which stalled for good whenever the
    // process's streams outnumbered its hardware queues and a wait shared a queue with the work it waited for.
    static const bool fuse_off = getenv("RTAMD_FUSE") && atoi(getenv("RTAMD_FUSE")) == 0; // experiment hook
    const bool fuse = !timed && !fuse_off; // (timed frames keep one kernel per launch, so that every kernel can be timed)
    const bool deep_bg = p->background_deep[0] != 0.0f || p->background_deep[1] != 0.0f || p->background_deep[2] != 0.0f;

    // Every pass starts with prep_kernel: it fills the pass's screen-coordinate tables and zeroes the bin counters; the first
    // one also zeroes the stage and work counters of the frame, which lie right in front of them.
    const size_t bin_clear = bins_ok ? ((256 + sizeof(int32_t) * (size_t)ws_frame.num_bins + 15) & ~(size_t)15) : faces_ok ? 256 : 0;
    // The per-pixel keys (indexed by local row and column) are zero between passes: the compaction kernel reads each once
    // and puts the zero back, and a fresh workspace is cleared when it is allocated.  Only the array shared by all callers
    // of the allocation-free entry point may hold anything.
    const size_t key_clear = faces_ok && plain ? (size_t)fs.ws.capacity * sizeof(unsigned long long) + (size_t)fs.ws.capacity / 64 * 4 : 0;
    int stage = stage_base;
    for (int xs = 0; xs < S; ++xs)
        for (int ys = 0; ys < S; ++ys) {
            fc.sub_x = (float)xs / S; // src/flyscene.cpp:816-817
            fc.sub_y = (float)ys / S;
            fc.accumulate = (xs | ys) != 0;
            // With the face pass planned, the pass's first launch also writes what a primary miss leaves into every pixel of
            // the output planes (whole lines, once), and the compaction kernel skips the tiles no face was accepted in;
            // a pass that ADDS to the planes (SSAA passes after the first) keeps the pixel-by-pixel form.
            ws_frame.prefilled = faces_ok && !fc.accumulate && sparse_view;
            // A frame alone on the GPU: the sweep rides along with the face pass (which does not depend on it) instead of
            // lengthening the chain of launches by its own duration (4K bunny alone: 0.171 -> 0.163 ms).  With frames in flight
            // the chain's length costs nothing and the separate, fully coalesced sweep is the cheaper one (2 % at 4K).
            static const int fill_forced = getenv("RTAMD_PREFILL_WITH_FACES") ? atoi(getenv("RTAMD_PREFILL_WITH_FACES")) : -1; // experiment hook
            ws_frame.prefill_with_faces = ws_frame.prefilled && fuse && (fill_forced >= 0 ? fill_forced != 0 : !ws_frame.full_waves);
            if (stage == 0) // (the first pass of the first band)
                launch_prep(fc, ws_frame, fs.block, kCounterHeadBytes + bin_clear, ws_frame.pix_key, key_clear, rgb, face_id, t_hit, stream);
            else
                launch_prep(fc, ws_frame, (char *)fs.block + kCounterHeadBytes, bin_clear, ws_frame.pix_key, key_clear, rgb, face_id, t_hit, stream);
            // Fused form: the shadow rays of level d share a launch with the closest hits of level d + 1, and the shading of
            // level d with the compaction of level d + 1 (each pair needs only what the launch before it produced), all on
            // the caller's stream.  The GPU runs only a few kernels at a time however many streams feed it, so what a frame
            // costs other frames in flight is the SUM of its launches' durations: pairing the long, mostly idle chain of a
            // bounce walk with the shadow rays beside it takes that sum down by a fifth, and a frame alone gets the overlap
            // the side streams below were made for.
            for (int d = 0; fuse && d < depths; ++d, ++stage) {
                const bool spawn = d + 1 < depths;
                if (d == 0) {
                    if (faces_ok) launch_primary_faces(count, s->ds, fc, ws_frame, s->info.num_device_faces, rgb, face_id, t_hit, stream);
                    if (bins_ok) launch_primary_bins(count, s->ds, fc, ws_frame, s->info.num_device_faces, /*cleared=*/true, stream);
                    // The tree walk behind them does something only where the pass gives way to it, which the device alone
                    // knows (the report of an earlier frame is a hint: it may belong to a frame still in flight, or to a view
                    // that has changed since): it is always launched, with a small grid where the hint says "kept".
                    launch_walk_closest(true, count, s->ds, fc, ws_frame, stage, 0, -1, stream);
                    launch_compact(true, s->ds, fc, ws_frame, stage, 0, spawn, false, rgb, face_id, t_hit, -1, stream, expect_hits[(size_t)stage]);
                }
                const int hits_here = expect_hits[(size_t)stage], rays_next = expect_next[(size_t)stage];
                if (spawn && can_fuse_shadow_walk(s->ds, fc)) {
                    launch_shadow_walk(count, s->ds, fc, ws_frame, stage, d, hits_here, rays_next, stream);
                } else {
                    if (nshadow > 0) launch_walk_shadow(count, s->ds, fc, ws_frame, stage, d, hits_here, stream);
                    if (spawn) launch_walk_closest(false, count, s->ds, fc, ws_frame, stage + 1, d + 1, rays_next, stream);
                }
                if (spawn && can_fuse_shade_compact(ws_frame)) {
                    launch_shade_compact(d == 0, s->ds, fc, ws_frame, stage, d, d + 2 < depths, deep_bg, rgb, hits_here, rays_next, stream);
                } else {
                    launch_shade(d == 0, s->ds, fc, ws_frame, stage, d, spawn, rgb, hits_here, stream);
                    if (spawn) launch_compact(false, s->ds, fc, ws_frame, stage + 1, d + 1, d + 2 < depths, deep_bg, rgb, face_id, t_hit, rays_next, stream);
                }
                if (deep_bg && d > 0) launch_misses(fc, ws_frame, stage, d, rgb, expect_next[(size_t)stage - 1], stream);
            }
            for (int d = 0; !fuse && d < depths; ++d, ++stage) {
                const bool spawn = d + 1 < depths;
                // ---- closest hits
                if ((rc = begin(d == 0 && (bins_ok || faces_ok) ? 6 : 0)) != RTAMD_OK) return rc;
                if (d == 0 && faces_ok) launch_primary_faces(count, s->ds, fc, ws_frame, s->info.num_device_faces, rgb, face_id, t_hit, stream);
                if (d == 0 && bins_ok) launch_primary_bins(count, s->ds, fc, ws_frame, s->info.num_device_faces, /*cleared=*/true, stream);
                launch_walk_closest(d == 0, count, s->ds, fc, ws_frame, stage, d, d > 0 ? expect_next[(size_t)stage - 1] : -1, stream); // after the bins: only on their overflow
                if ((rc = end()) != RTAMD_OK) return rc;
                if ((rc = begin(2)) != RTAMD_OK) return rc;
                launch_compact(d == 0, s->ds, fc, ws_frame, stage, d, spawn, deep_bg && d > 0, rgb, face_id, t_hit, d > 0 ? expect_next[(size_t)stage - 1] : -1, stream,
                               d == 0 ? expect_hits[(size_t)stage] : -1);
                if ((rc = end()) != RTAMD_OK) return rc;
                // ---- shadow rays
                if (nshadow > 0) {
                    if ((rc = begin(1)) != RTAMD_OK) return rc;
                    launch_walk_shadow(count, s->ds, fc, ws_frame, stage, d, expect_hits[(size_t)stage], stream);
                    if ((rc = end()) != RTAMD_OK) return rc;
                }
                // ---- shading, in depth order
                if ((rc = begin(3)) != RTAMD_OK) return rc;
                launch_shade(d == 0, s->ds, fc, ws_frame, stage, d, spawn, rgb, expect_hits[(size_t)stage], stream);
                if (deep_bg && d > 0) launch_misses(fc, ws_frame, stage, d, rgb, expect_next[(size_t)stage - 1], stream);
                if ((rc = end()) != RTAMD_OK) return rc;
            }
        }
    if (S > 1) {
        if ((rc = begin(4)) != RTAMD_OK) return rc;
        launch_scale(rgb, pixels * 3, (float)(S * S), stream);
        if ((rc = end()) != RTAMD_OK) return rc;
    }
    if (timed && band == bands - 1) HIP_TRY(hipEventRecord(s->events[s->open_frame_event].stop, stream));
    if (fs.done && band == bands - 1) HIP_TRY(hipEventRecord(fs.done, stream));
    fs.used = true;
    HIP_TRY(hipGetLastError());

    s->last = *whole;
    s->last_stages = stage_base + stages;
    s->last_depths = depths;
    s->last_primary = (band ? s->last_primary : 0) + (uint64_t)pixels * (uint64_t)(S * S);
    s->last_stream = stream;
    s->last_frame_slot = &fs;
    s->have_frame = true;
    return RTAMD_OK;
}

} // namespace

extern "C" {

int rtamd_render_device(rtamd_scene *s, const rtamd_camera *cam, const rtamd_render_params *p, float *rgb, int32_t *face_id, float *t_hit,
                        void *hip_stream) {
    int rc = check_params(s, cam, p);
    if (rc != RTAMD_OK) return rc;
    if (!rgb) return fail(RTAMD_ERR_INVALID, "null rgb output");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != s->device) return fail(RTAMD_ERR_INVALID, "scene lives on device %d but device %d is current", s->device, dev);
    hipStream_t stream = (hipStream_t)hip_stream;
    FrameShape shape;
    if ((rc = frame_shape(s, p, shape)) != RTAMD_OK) return rc;
    s->have_frame = false;
    if (shape.pixels == 0) return RTAMD_OK;
    FrameSlot *slot = nullptr;
    if ((rc = acquire_slot(s, stream, &slot)) != RTAMD_OK) return rc;
    const int bands = plan_bands(s, p, shape, (size_t)s->workspace_limit, s->bin_primary ? (size_t)shape.bins_x * (size_t)shape.bins_y : 0, s->face_primary);
    if (bands == 1) {
        rc = ensure_workspace(s, *slot, shape.slots, (size_t)shape.bins_x * (size_t)shape.bins_y, shape.depths);
        if (rc != RTAMD_OK) return rc;
        return enqueue_frame(s, *slot, shape, cam, p, rgb, face_id, t_hit, stream, /*plain=*/false);
    }
    // In bands: band b is part b of `bands` under row tiles of one band's height, i.e. the image's rows
    // [b * band_rows, (b + 1) * band_rows), rendered like one rank's share of a partitioned frame and written where those
    // rows lie in the caller's planes.  The bands follow each other on the stream through the same work buffers.
    rtamd_render_params pb = *p;
    pb.row_tile = band_height(p->height, bands), pb.num_parts = bands, pb.part = 0;
    FrameShape band_shape;
    if ((rc = frame_shape(s, &pb, band_shape)) != RTAMD_OK) return rc;
    rc = ensure_workspace(s, *slot, band_shape.slots, (size_t)band_shape.bins_x * (size_t)band_shape.bins_y, band_shape.depths);
    if (rc != RTAMD_OK) return rc;
    for (int b = 0; b < bands; ++b) {
        pb.part = b;
        if ((rc = frame_shape(s, &pb, band_shape)) != RTAMD_OK) return rc;
        const size_t at = (size_t)b * (size_t)pb.row_tile * (size_t)p->width; // first pixel of the band
        rc = enqueue_frame(s, *slot, band_shape, cam, &pb, rgb + 3 * at, face_id ? face_id + at : nullptr, t_hit ? t_hit + at : nullptr, stream,
                           /*plain=*/false, b, bands, p);
        if (rc != RTAMD_OK) return rc;
    }
    return RTAMD_OK;
}

int rtamd_render_device_noalloc(rtamd_scene *s, const rtamd_camera *cam, const rtamd_render_params *p, float *rgb, int32_t *face_id,
                                float *t_hit, void *hip_stream) {
    int rc = check_params(s, cam, p);
    if (rc != RTAMD_OK) return rc;
    if (!rgb) return fail(RTAMD_ERR_INVALID, "null rgb output");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != s->device) return fail(RTAMD_ERR_INVALID, "scene lives on device %d but device %d is current", s->device, dev);
    FrameShape shape;
    if ((rc = frame_shape(s, p, shape)) != RTAMD_OK) return rc;
    s->have_frame = false;
    if (shape.pixels == 0) return RTAMD_OK;
    // Work buffers: laid out in the device's static array instead of a hipMalloc'ed block.  A frame that does not fit whole
    // is rendered in bands of rows (as rtamd_render_device does under its workspace limit), each band through the array.
    const size_t ntris = (size_t)s->info.num_device_faces;
    Workspace sizes{};
    const int bands = plan_bands(s, p, shape, kScratchBytes, 0, false);
    rtamd_render_params pb = *p;
    FrameShape first = shape; // the largest band (or the whole frame)
    if (bands > 1) {
        pb.row_tile = band_height(p->height, bands), pb.num_parts = bands, pb.part = 0;
        if ((rc = frame_shape(s, &pb, first)) != RTAMD_OK) return rc;
    }
    const size_t need_plain = layout_workspace(nullptr, first.slots, first.nshadow, 0, ntris, false, first.depths, sizes);
    if (need_plain > kScratchBytes)
        return fail(RTAMD_ERR_LIMIT, "a frame of %d x %d pixels with %zu shadow rays per hit needs %zu bytes of work buffers%s; "
                    "rtamd_render_device_noalloc holds %zu (use rtamd_render_device)", p->width, shape.local_rows, shape.nshadow, need_plain,
                    bands > 1 ? " per band of rows, in as many bands as its passes allow" : "", kScratchBytes);
    // the buffers of the screen-space pass for the primary rays (per-pixel keys, or bins) only when they fit as well
    // (otherwise those rays walk the tree: same image)
    const size_t bins = s->bin_primary ? (size_t)first.bins_x * (size_t)first.bins_y : 0;
    const bool with_keys = s->face_primary && layout_workspace(nullptr, first.slots, first.nshadow, 0, ntris, true, first.depths, sizes) <= kScratchBytes;
    const bool with_bins = bins > 0 && layout_workspace(nullptr, first.slots, first.nshadow, bins, ntris, false, first.depths, sizes) <= kScratchBytes;
    char *base = s->scratch; // looked up when the scene was created
    if (!base && (rc = scratch_pointer(dev, &base)) != RTAMD_OK) return rc;
    FrameSlot &fs = s->static_slot;
    layout_workspace(base, first.slots, first.nshadow, with_bins ? bins : 0, ntris, with_keys, first.depths, fs.ws);
    fs.block = base;
    fs.bins = with_bins ? bins : 0;
    fs.stream = (hipStream_t)hip_stream;
    if (bands == 1) return enqueue_frame(s, fs, shape, cam, p, rgb, face_id, t_hit, (hipStream_t)hip_stream, /*plain=*/true);
    for (int b = 0; b < bands; ++b) {
        pb.part = b;
        FrameShape band_shape;
        if ((rc = frame_shape(s, &pb, band_shape)) != RTAMD_OK) return rc;
        const size_t at = (size_t)b * (size_t)pb.row_tile * (size_t)p->width; // first pixel of the band
        rc = enqueue_frame(s, fs, band_shape, cam, &pb, rgb + 3 * at, face_id ? face_id + at : nullptr, t_hit ? t_hit + at : nullptr,
                           (hipStream_t)hip_stream, /*plain=*/true, b, bands, p);
        if (rc != RTAMD_OK) return rc;
    }
    return RTAMD_OK;
}


int rtamd_scene_set_workspace_limit(rtamd_scene *s, uint64_t bytes) {
    if (!s) return fail(RTAMD_ERR_INVALID, "null scene");
    s->workspace_limit = bytes;
    return RTAMD_OK;
}

int rtamd_scene_workspace_bytes(const rtamd_scene *s, uint64_t *bytes) {
    if (!s || !bytes) return fail(RTAMD_ERR_INVALID, "null scene or output");
    uint64_t total = 0;
    for (int i = 0; i < s->num_slots; ++i) total += s->slots[i].bytes;
    *bytes = total + (uint64_t)s->staging_pixels * 16;
    return RTAMD_OK;
}

int rtamd_render_stats(rtamd_scene *s, rtamd_stats *st) {
    if (!s || !st) return fail(RTAMD_ERR_INVALID, "null scene or stats");
    std::memset(st, 0, sizeof *st);
    if (!s->have_frame) return RTAMD_OK;
    for (int i = 0; i < s->num_slots; ++i) // timing events of every frame in flight must have completed
        if (s->slots[i].used) HIP_TRY(hipEventSynchronize(s->slots[i].done));
    if (s->last_frame_slot == &s->static_slot) HIP_TRY(hipStreamSynchronize(s->last_stream)); // that slot has no event
    const Workspace &ws_last = s->last_frame_slot->ws;
    std::vector<StageCounters> sc((size_t)kMaxStages);
    WorkCounters wc[3];
    HIP_TRY(hipMemcpy(sc.data(), ws_last.stage, sizeof(StageCounters) * kMaxStages, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(wc, ws_last.work, sizeof wc, hipMemcpyDeviceToHost));
    st->rays_primary = s->last_primary;
    for (int i = 0; i < s->last_stages; ++i) {
        st->hits += (uint64_t)sc[i].hits;
        st->rays_bounce += (uint64_t)sc[i].next;
    }
    // shadow rays: what the shadow-ray kernels counted while they shot them (not hits x lights x samples)
    unsigned long long shot[kShotSlots * kShotStride];
    HIP_TRY(hipMemcpy(shot, ws_last.shadow_shot, sizeof shot, hipMemcpyDeviceToHost));
    for (int i = 0; i < kShotSlots; ++i) st->rays_shadow += shot[i * kShotStride];
    for (int k = 0; k < 3; ++k) {
        st->box_tests[k] = wc[k].box_tests;
        st->tri_plane[k] = wc[k].tri_plane;
        st->tri_edge[k] = wc[k].tri_edge;
    }
    if (s->last_frame_slot->rastered) {
        BinState bs;
        HIP_TRY(hipMemcpy(&bs, ws_last.bin_state, sizeof bs, hipMemcpyDeviceToHost));
        st->primary_path = !bs.fallback && bs.num_big <= kMaxHugeFaceList ? 1 : 0;
    }
    if (s->last_frame_slot->binned) {
        BinState bs;
        HIP_TRY(hipMemcpy(&bs, ws_last.bin_state, sizeof bs, hipMemcpyDeviceToHost));
        st->bin_entries = (uint32_t)std::min<unsigned long long>(bs.total, 0xffffffffull);
        st->bins_nonempty = (uint32_t)bs.nonempty;
        st->bin_overflow = bs.fallback;
        const long long limit = s->bin_always ? 0x7fffffffll : (long long)kMaxAverageBinList;
        st->primary_path = !bs.fallback && (long long)bs.num_big <= 4 * limit && bs.total <= (unsigned long long)(limit * (long long)std::max(bs.nonempty, 1)) ? 1 : 0;
    }
    for (size_t i = 0; i < s->events_used; ++i) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, s->events[i].start, s->events[i].stop));
        if (s->events[i].kind == 0) st->ms_trace += ms, ++st->launches_trace;
        if (s->events[i].kind == 1) st->ms_shadow += ms, ++st->launches_shadow;
        if (s->events[i].kind == 2) st->ms_compact += ms;
        if (s->events[i].kind == 3) st->ms_shade += ms;
        if (s->events[i].kind == 5) st->ms_frames += ms, ++st->frames;
        if (s->events[i].kind == 6) st->ms_raster += ms, ++st->launches_raster;
    }
    s->events_used = 0;
    return RTAMD_OK;
}

int rtamd_host_alloc(void **out, uint64_t bytes) {
    if (!out) return fail(RTAMD_ERR_INVALID, "null output pointer");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? (size_t)bytes : 64, hipHostMallocDefault));
    return RTAMD_OK;
}

int rtamd_host_free(void *ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return RTAMD_OK;
}

int rtamd_render(rtamd_scene *s, const rtamd_camera *cam, const rtamd_render_params *p, float *rgb, int32_t *face_id, rtamd_stats *stats) {
    int rc = check_params(s, cam, p);
    if (rc != RTAMD_OK) return rc;
    if (!rgb) return fail(RTAMD_ERR_INVALID, "null rgb output");
    const size_t pixels = (size_t)rtamd_local_rows(p) * (size_t)p->width;
    if (pixels == 0) {
        if (stats) std::memset(stats, 0, sizeof *stats);
        return RTAMD_OK;
    }
    if (s->staging_pixels < pixels) {
        if (s->d_rgb) HIP_TRY(hipFree(s->d_rgb));
        if (s->d_face) HIP_TRY(hipFree(s->d_face));
        s->d_rgb = nullptr, s->d_face = nullptr, s->staging_pixels = 0;
        HIP_TRY(hipMalloc((void **)&s->d_rgb, pixels * 12));
        HIP_TRY(hipMalloc((void **)&s->d_face, pixels * 4));
        s->staging_pixels = pixels;
    }
    // Download: one asynchronous copy per plane behind the frame, one wait.  Into the caller's buffer when that is
    // page-locked (rtamd_host_alloc, hipHostMalloc, hipHostRegister: the copy then runs at the link's rate), otherwise
    // into page-locked buffers of the scene and from there with a few host threads (a pageable hipMemcpy stages
    // through small bounce buffers and was four fifths of this call).
    auto pinned = [](const void *ptr) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
            (void)hipGetLastError(); // an ordinary host pointer is reported as an error
            return false;
        }
        return attr.type == hipMemoryTypeHost;
    };
    const bool rgb_direct = pinned(rgb), face_direct = face_id && pinned(face_id);
    const size_t need = (rgb_direct ? 0 : pixels * 12) + (face_id && !face_direct ? pixels * 4 : 0);
    if (s->h_staging_bytes < need) {
        if (s->h_staging) HIP_TRY(hipHostFree(s->h_staging));
        s->h_staging = nullptr, s->h_staging_bytes = 0;
        HIP_TRY(hipHostMalloc(&s->h_staging, need, hipHostMallocDefault));
        s->h_staging_bytes = need;
    }
    char *const stage_rgb = (char *)s->h_staging, *const stage_face = stage_rgb + (rgb_direct ? 0 : pixels * 12);
    rc = rtamd_render_device(s, cam, p, s->d_rgb, face_id ? s->d_face : nullptr, nullptr, nullptr);
    if (rc != RTAMD_OK) return rc;
    HIP_TRY(hipMemcpyAsync(rgb_direct ? (void *)rgb : (void *)stage_rgb, s->d_rgb, pixels * 12, hipMemcpyDeviceToHost, nullptr));
    if (face_id) HIP_TRY(hipMemcpyAsync(face_direct ? (void *)face_id : (void *)stage_face, s->d_face, pixels * 4, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    auto host_copy = [](void *dst, const void *src, size_t bytes) {
        const size_t workers = bytes >= ((size_t)4 << 20) ? 4 : 1;
        if (workers == 1) {
            std::memcpy(dst, src, bytes);
            return;
        }
        std::vector<std::thread> pool;
        const size_t chunk = ((bytes / workers) + 4095) & ~(size_t)4095;
        for (size_t w = 0; w < workers; ++w) {
            const size_t a = std::min(bytes, w * chunk), b = std::min(bytes, (w + 1) * chunk);
            if (b > a) pool.emplace_back([=] { std::memcpy((char *)dst + a, (const char *)src + a, b - a); });
        }
        for (std::thread &t : pool) t.join();
    };
    if (!rgb_direct) host_copy(rgb, stage_rgb, pixels * 12);
    if (face_id && !face_direct) host_copy(face_id, stage_face, pixels * 4);
    if (stats) return rtamd_render_stats(s, stats);
    return RTAMD_OK;
}

int rtamd_render_ppm(rtamd_scene *s, const rtamd_camera *cam, const rtamd_render_params *p, uint8_t *pixels, int32_t *exact, rtamd_stats *stats) {
    int rc = check_params(s, cam, p);
    if (rc != RTAMD_OK) return rc;
    if (!pixels) return fail(RTAMD_ERR_INVALID, "null pixel output");
    if (exact) *exact = 1;
    const size_t npix = (size_t)rtamd_local_rows(p) * (size_t)p->width;
    if (npix == 0) {
        if (stats) std::memset(stats, 0, sizeof *stats);
        return RTAMD_OK;
    }
    if (s->staging_pixels < npix) {
        if (s->d_rgb) HIP_TRY(hipFree(s->d_rgb));
        if (s->d_face) HIP_TRY(hipFree(s->d_face));
        s->d_rgb = nullptr, s->d_face = nullptr, s->staging_pixels = 0;
        HIP_TRY(hipMalloc((void **)&s->d_rgb, npix * 12));
        HIP_TRY(hipMalloc((void **)&s->d_face, npix * 4));
        s->staging_pixels = npix;
    }
    const size_t nbytes = (npix * 3 + 15) & ~(size_t)15; // the flag sits behind the image, 16-byte aligned: one download brings both
    if (s->bytes_pixels < npix) {
        if (s->d_bytes) HIP_TRY(hipFree(s->d_bytes));
        s->d_bytes = nullptr, s->bytes_pixels = 0;
        HIP_TRY(hipMalloc((void **)&s->d_bytes, nbytes + 16));
        s->bytes_pixels = npix;
    }
    hipPointerAttribute_t attr;
    bool direct = hipPointerGetAttributes(&attr, pixels) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!direct) (void)hipGetLastError(); // an ordinary host pointer is reported as an error
    const size_t need = (direct ? 0 : nbytes) + 16;
    if (s->h_staging_bytes < need) {
        if (s->h_staging) HIP_TRY(hipHostFree(s->h_staging));
        s->h_staging = nullptr, s->h_staging_bytes = 0;
        HIP_TRY(hipHostMalloc(&s->h_staging, need, hipHostMallocDefault));
        s->h_staging_bytes = need;
    }
    int *const d_flag = (int *)(s->d_bytes + nbytes);
    volatile int *const h_flag = (volatile int *)s->h_staging;
    uint8_t *const h_pixels = direct ? pixels : (uint8_t *)s->h_staging + 16;
    HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int), nullptr));
    rc = rtamd_render_device(s, cam, p, s->d_rgb, nullptr, nullptr, nullptr);
    if (rc != RTAMD_OK) return rc;
    launch_quantize_checked(s->d_rgb, s->d_bytes, npix * 3, d_flag, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_pixels, s->d_bytes, npix * 3, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync((void *)h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    if (!direct) std::memcpy(pixels, h_pixels, npix * 3);
    if (exact) *exact = *h_flag ? 0 : 1;
    if (stats) return rtamd_render_stats(s, stats);
    return RTAMD_OK;
}

int rtamd_quantize_ppm(const float *rgb_dev, uint8_t *bytes_dev, uint64_t count, void *hip_stream) {
    if (count == 0) return RTAMD_OK;
    if (!rgb_dev || !bytes_dev) return fail(RTAMD_ERR_INVALID, "null argument");
    if (((uintptr_t)rgb_dev & 15) != 0 || ((uintptr_t)bytes_dev & 3) != 0) return fail(RTAMD_ERR_INVALID, "rgb must be 16-byte aligned, bytes 4-byte aligned");
    launch_quantize(rgb_dev, bytes_dev, (size_t)count, (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return RTAMD_OK;
}

int rtamd_assemble_rows(const rtamd_render_params *p, const void *src, void *dst, int32_t channels, void *hip_stream) {
    if (!p || !src || !dst) return fail(RTAMD_ERR_INVALID, "null argument");
    if (p->width <= 0 || p->height <= 0 || channels <= 0 || p->row_tile < 1 || p->num_parts < 1)
        return fail(RTAMD_ERR_INVALID, "bad image or partition");
    launch_assemble(src, dst, p->width, p->height, channels, p->row_tile, p->num_parts, (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return RTAMD_OK;
}

} // extern "C"
