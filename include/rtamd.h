/*
 * rtamd.h -- C ABI of the MI355X (gfx950) render pass for the Flyscene ray tracer.
 *
 * The reference renderer has no plugin or FFI boundary: Flyscene::raytraceScene()
 * (src/flyscene.cpp:381-459) reads the members of one C++ object and fans out over std::thread.
 * This header is the boundary a maintainer binds instead.  It is cut where the reference
 * spawns and joins its worker threads (src/flyscene.cpp:413-435): everything tracePixels /
 * traceRay read crosses it as plain arrays, and the image comes back in the layout of
 * `pixel_data[y][x]` (src/flyscene.cpp:399-403,830).
 *
 * Conventions: every function returns RTAMD_OK (0) or an error code; the text of the last error
 * of the calling thread is returned by rtamd_last_error().  Input arrays are copied; the caller
 * keeps ownership of everything it passes.  A scene lives on the HIP device that is current when
 * it is created; calls on one scene must not overlap.  All arithmetic is binary32 as in the
 * reference (Eigen::Vector3f).
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTAMD_OK 0
#define RTAMD_ERR_INVALID 1   /* bad argument (null pointer, index out of range, size <= 0)   */
#define RTAMD_ERR_HIP 2       /* a HIP runtime call failed                                      */
#define RTAMD_ERR_LIMIT 3     /* scene or image exceeds a documented limit                     */
#define RTAMD_ERR_NO_DEVICE 4 /* no usable gfx950 device                                        */

/* Face-id plane values besides face indices. */
#define RTAMD_FACE_SPHERE (-1) /* src/flyscene.cpp:721 */
#define RTAMD_FACE_MISS (-2)

typedef struct rtamd_scene rtamd_scene;

/* Tucano::Material::Mtl getters read on the path (materials/mtl.hpp:118-121). */
typedef struct {
    float ka[3];
    float kd[3];
    float ks[3];
    float shininess;
} rtamd_material;

/* struct Sphere, src/defs.h:22-25. */
typedef struct {
    float center[3];
    float radius;
} rtamd_sphere;

/* One entry of Flyscene::texturedatas (src/flyscene.hpp:187, filled at src/flyscene.cpp:52-71): the
 * texels of a material's map_kd image as ImageImporter::loadPPMImageData returns them
 * (utils/ppmIO.hpp:36-86): value / maximum, rows flipped, so row 0 is the LAST row of the file. */
typedef struct {
    int32_t width, height;
    const float *rgb; /* width*height*3 */
} rtamd_texture;

/* struct PrecomputedData (src/defs.h:7-20) plus what shadeOffFace/getLightIntensity read beside
 * it: materials (src/flyscene.hpp:181), lights (src/flyscene.hpp:163) and spheres (:166). */
typedef struct {
    const float *vertices;             /* num_vertices*3, world space  (PrecomputedData::vertices)  */
    const float *normals;              /* num_vertices*3, unit          (PrecomputedData::normals)   */
    int32_t num_vertices;
    const int32_t *face_vertex_ids;    /* num_faces*3                   (faceVertexIds)              */
    const float *face_normals;         /* num_faces*3, unit             (faceNormals)                */
    const float *face_origin_distance; /* num_faces                     (faceOriginDistance)         */
    const int32_t *face_material_ids;  /* num_faces, -1 = none          (faceMaterialIds)            */
    int32_t num_faces;
    const rtamd_material *materials;
    int32_t num_materials;
    const float *light_positions;      /* num_lights*3                  (Flyscene::lights)           */
    const float *light_colors;         /* num_lights*3 (carried; the reference never applies it)     */
    int32_t num_lights;
    const float *light_samples;        /* (num_lights - num_unsampled_lights)*samples_per_light*3
                                          (PrecomputedData::lights)                                  */
    int32_t samples_per_light;
    const rtamd_sphere *spheres;
    int32_t num_spheres;
    /* Texture branch of shadeOffFace (src/flyscene.cpp:476-481,516-542); all three may be NULL / 0. */
    const float *face_texcoords;       /* num_faces*6: (u,v) of the three corners, Mesh::Face::texcoord  */
    const int32_t *material_textures;  /* num_materials: index into textures, -1 = material has no map_kd */
    const rtamd_texture *textures;
    int32_t num_textures;
    /* The LAST num_unsampled_lights entries of the light list have no shadow sample points: the state
     * Flyscene::addLight leaves (src/flyscene.hpp:76 appends to `lights`; precomputeLights ran once, in
     * initialize, src/flyscene.cpp:117).  shadeOffFace loops over all num_lights (:505), getLightIntensity
     * over the sampled ones only (:880).  0 = every light is sampled. */
    int32_t num_unsampled_lights;
} rtamd_scene_desc;

#define RTAMD_PRIMARY_RAYS_AUTO 0 /* all primary rays share the eye: every face is projected and tested against the pixels of
                                     its own box (the pixels keep the nearest through an atomic maximum); the BVH where
                                     that is cheaper (many faces per pixel, or the eye inside the mesh)            */
#define RTAMD_PRIMARY_RAYS_TREE 1 /* primary rays walk the BVH like bounce rays                            */
#define RTAMD_PRIMARY_RAYS_BINS 2 /* per-pass screen binning of the faces into 4x4-pixel bins, every pixel running through its
                                     bin's list; kept whenever possible, also where the lists come out long         */
#define RTAMD_SHADOW_RAYS_AUTO 0 /* per-sample candidate grids when every sample direction allows it   */
#define RTAMD_SHADOW_RAYS_TREE 1 /* shadow rays walk the BVH like every other ray                       */

/* Who builds the hierarchy that replaces boundingBox::splitBox (src/boundingBox.h:69-194).  Images do not
 * depend on the choice; only the time to build and the time per frame do. */
#define RTAMD_BUILDER_HOST_SAH 0    /* binned surface-area heuristic on the host: best tree, seconds for 10^6 faces */
#define RTAMD_BUILDER_DEVICE_LBVH 1 /* Morton-order radix tree and the shadow-ray grids built on the GPU: milliseconds,
                                       somewhat slower bounce rays;
                                       falls back to the host builder when it does not apply (rtamd_scene_info::builder) */

typedef struct {
    int32_t max_leaf_faces; /* 1..15; 0 = default (4)                                              */
    int32_t sah_bins;       /* 4..64; 0 = default (16)                                             */
    int32_t shadow_rays;    /* RTAMD_SHADOW_RAYS_*                                                 */
    int32_t primary_rays;   /* RTAMD_PRIMARY_RAYS_*                                                */
    int32_t builder;        /* RTAMD_BUILDER_*                                                     */
    int32_t reserved[3];
} rtamd_build_params;

/* What tracePixels/screenToWorld read from Tucano::Camera (camera.hpp:84,118-121,158-181). */
typedef struct {
    float origin[3];    /* getCenter()                                                           */
    float inv_view[12]; /* inverseViewMatrix, row-major 3x4 = [linear | translation]             */
    float viewport[4];  /* (x, y, width, height)                                                 */
    float aspect_ratio; /* as given to setPerspectiveMatrix (src/flyscene.cpp:35)                */
    float fovy;         /* degrees                                                               */
} rtamd_camera;

#define RTAMD_FLAG_COUNT_WORK 1u /* also count box tests / triangle tests (slower kernels)        */
#define RTAMD_FLAG_TIME_KERNELS 2u /* bracket every launch with HIP events on the render stream; such a frame runs one
                                      kernel per launch (untimed frames pair independent stages in one launch)        */
/* Kept for callers that set it: every frame runs on the caller's stream alone (the library has no streams of its own),
 * so the flag changes nothing. */
#define RTAMD_FLAG_SINGLE_STREAM 4u

/* The compile-time macros of src/flyscene.hpp:22-41 and the literals of src/flyscene.cpp as
 * run-time fields, plus the row-tile partition used for multi-GPU rendering. */
typedef struct {
    int32_t width, height;    /* full image; must equal the viewport size (src/flyscene.cpp:394)   */
    int32_t ssaa_level;       /* SSAALEVEL                                                         */
    int32_t max_reflections;  /* MAXREFLECTIONS                                                    */
    float reflect_threshold;  /* 0.25, src/flyscene.cpp:631                                        */
    float bounce_offset;      /* 0.01, src/flyscene.cpp:636                                        */
    float t_epsilon;          /* 1e-5, src/flyscene.cpp:707,760                                    */
    float background[3];      /* 0.4,  src/flyscene.cpp:653-658                                    */
    float background_deep[3]; /* 0.0,  src/flyscene.cpp:661-665                                    */
    /* Row-tile partition: the image is cut into tiles of `row_tile` rows; tile k belongs to part
     * k % num_parts.  A render call produces only the rows of `part`, packed top to bottom.
     * (row_tile, 1, 0) renders the whole image.  The reference's analogue is the row interleave
     * of its threads, src/flyscene.cpp:809. */
    int32_t row_tile, num_parts, part;
    uint32_t flags;
    int32_t reserved[4];
} rtamd_render_params;

typedef struct {
    /* Counters of the most recent frame. */
    uint64_t rays_primary; /* one per pixel sub-sample                                            */
    uint64_t rays_shadow;  /* shadow rays shot, counted by the kernels that shoot them: one per hit, sampled light and
                              sample point (src/flyscene.cpp:880-887)                              */
    uint64_t rays_bounce;  /* reflection rays (src/flyscene.cpp:636)                              */
    uint64_t hits;         /* closest-hit queries that found a surface                            */
    /* RTAMD_FLAG_COUNT_WORK only, most recent frame; [0] = BVH walk for closest hits (bounce rays, and
     * primary rays when they are not binned), [1] = shadow-ray kernel (candidate grids or BVH walk),
     * [2] = screen-bin pass of the primary rays: */
    uint64_t box_tests[3]; /* ray/AABB slab tests                                                 */
    uint64_t tri_plane[3]; /* plane-distance evaluations                                          */
    uint64_t tri_edge[3];  /* faces that reached the edge tests                                   */
    /* RTAMD_FLAG_TIME_KERNELS only: HIP-event durations in milliseconds, summed over every launch
     * and frame enqueued since the previous rtamd_render_stats call: */
    float ms_trace;        /* closest-hit BVH walk kernel                                         */
    float ms_shadow;       /* shadow-ray kernel                                                   */
    float ms_compact;      /* hit-list compaction kernel                                          */
    float ms_shade;        /* shading kernel                                                      */
    float ms_frames;       /* first launch to last launch of each frame, summed                   */
    int32_t launches_trace;
    int32_t launches_shadow;
    int32_t frames;
    float ms_raster;       /* screen-space pass of the primary rays: the face raster kernel, or with forced bins the
                            * bin count, offsets, fill and raster kernels                          */
    int32_t launches_raster; /* passes, not kernels                                               */
    /* LAST pass (SSAA sub-sample) and last band of rows of the most recent frame -- earlier passes and bands may have gone the
     * other way; the bin figures only when it binned its faces (0 otherwise): */
    uint32_t bin_entries;    /* face references in the screen bins                                */
    uint32_t bins_nonempty;  /* bins holding at least one face                                    */
    int32_t primary_path;    /* 1: primary rays were resolved in screen space (face by face, or through the bins),
                              * 0: by the BVH walk (not attempted, buffers too small, or too costly for the view) */
    int32_t bin_overflow;    /* 1: the bin buffer was too small for the view                      */
} rtamd_stats;

typedef struct {
    int32_t num_faces;        /* faces given                                                      */
    int32_t num_device_faces; /* faces in the device BVH (zero-area faces are left out)           */
    int32_t num_nodes;        /* 128-byte inner nodes (up to four children each)                  */
    int32_t num_leaves;
    int32_t depth;
    int32_t device;           /* HIP device ordinal                                               */
    uint64_t device_bytes;    /* scene data resident in HBM                                       */
    float build_ms;           /* host BVH (and shadow grid) build                                 */
    float upload_ms;
    uint64_t shadow_grid_entries; /* face references in the shadow-ray grids, 0 = tree walk        */
    int32_t builder;          /* RTAMD_BUILDER_* that produced the hierarchy in use               */
    float hierarchy_ms;       /* part of build_ms spent on the hierarchy alone                    */
} rtamd_scene_info;

const char *rtamd_last_error(void);

/* Fill the defaults of the reference: Flycamera::reset + Flyscene::initialize(w, h)
 * (utils/flycamera.hpp:76-86,166-191; src/flyscene.cpp:35-36) and the macros above. */
void rtamd_camera_default(rtamd_camera *cam, int32_t width, int32_t height);
void rtamd_render_params_default(rtamd_render_params *p, int32_t width, int32_t height);

/* Replaces the tail of Flyscene::initialize (src/flyscene.cpp:116-134): takes the flattened scene,
 * builds the four-way BVH that stands in for boundingBox::splitBox (src/boundingBox.h:69-194)
 * and uploads everything to the current HIP device. */
int rtamd_scene_create(const rtamd_scene_desc *desc, const rtamd_build_params *build, rtamd_scene **out);
int rtamd_scene_destroy(rtamd_scene *scene);
/* Replaces the scene's light list (Flyscene::lights, src/flyscene.hpp:163) and leaves the shadow sample points
 * (PrecomputedData::lights) as they are: what Flyscene::addLight does to a scene that is already initialised
 * (src/flyscene.hpp:76).  num_lights must not be smaller than the number of sampled lights the scene was created
 * with; the lights beyond those are shaded (src/flyscene.cpp:505) but cast no shadow rays (:880).  Waits for the
 * frames in flight; nothing is rebuilt. */
int rtamd_scene_set_lights(rtamd_scene *scene, const float *light_positions, const float *light_colors, int32_t num_lights);
int rtamd_scene_get_info(const rtamd_scene *scene, rtamd_scene_info *info);
/* Device memory the scene holds beside its own data (rtamd_scene_info::device_bytes): the work buffers of the frames in
 * flight, one set per stream it has been rendered on (hit lists, ray queues, shadow results, per-pixel keys), sized for the
 * largest image and deepest recursion seen so far, plus the staging of rtamd_render.  A frame without reflections takes
 * about 40 + lights * samples_per_light bytes per pixel, one bounce about 130, deeper recursion about 210. */
int rtamd_scene_workspace_bytes(const rtamd_scene *scene, uint64_t *bytes);
/* Bound on the work buffers of ONE frame in flight (default 2 GiB; 0 = none).  An unpartitioned frame (num_parts == 1)
 * whose buffers would exceed it is rendered in bands of rows, one after the other on the caller's stream through buffers
 * of one band's size: the same image and the same counters in rtamd_stats, bit for bit (each band is rendered like one
 * part of a partitioned frame, src/flyscene.cpp:808-809).  A 7680x4320 frame with eight lights and four bounces takes
 * 6.7 GB whole and 1.7 GB in four bands.  Partitioned frames, and frames with so many passes that the bands would run out
 * of stage counters (bands * ssaa^2 * (max_reflections + 1) <= 64), keep fewer bands than the bound asks for.
 * No counterpart in the reference, whose working set per thread is a call stack. */
int rtamd_scene_set_workspace_limit(rtamd_scene *scene, uint64_t bytes);

/* Host only (no device needed): rank[f] = position of face f in the order in which the reference's
 * triangleIntersection meets the faces (leaves of boundingBox::splitBox in lower/upper preorder,
 * src/boundingBox.h:69-194,295-307; faces in leaf order, src/flyscene.cpp:750-751).  The kernels use
 * it to reproduce the reference's rule for equal hit distances: the face met later wins
 * (src/flyscene.cpp:760).  rank = desc->num_faces ints. */
int rtamd_reference_face_rank(const rtamd_scene_desc *desc, int32_t *rank);

/* Number of image rows part `p->part` renders, and the image row of local row `local_row`. */
int32_t rtamd_local_rows(const rtamd_render_params *p);
int32_t rtamd_global_row(const rtamd_render_params *p, int32_t local_row);

/* Replaces spawn..join of Flyscene::raytraceScene (src/flyscene.cpp:413-435) together with
 * tracePixels/traceRay/intersects/getLightIntensity/shadeOffFace below it.  Blocking.  Host
 * buffers: rgb = local_rows*width*3 floats, row 0 = top, i.e. pixel_data[y][x] (:399-403,830);
 * face_id (may be NULL) = local_rows*width ints: primary hit face, RTAMD_FACE_SPHERE or
 * RTAMD_FACE_MISS; stats may be NULL. */
int rtamd_render(rtamd_scene *scene, const rtamd_camera *cam, const rtamd_render_params *p, float *rgb,
                 int32_t *face_id, rtamd_stats *stats);

/* The frame as Flyscene::raytraceScene leaves it: not pixel_data but what writePPMImage makes of it
 * (src/flyscene.cpp:445; utils/ppmIO.hpp:203), pixels[i] = min(255, (int)(255 * rgb[i])), local_rows*width*3 bytes, quantised
 * on the device so that a quarter of the float framebuffer's bytes cross the host link.  Blocking, host buffer (page-locked
 * or not, as for rtamd_render).  *exact (may be NULL) = 1 when the bytes are what the reference prints for every channel;
 * 0 when some channel is <= -1/255, not a number or beyond the int range -- the reference prints a negative number or is
 * undefined there -- and the caller who cares renders that frame with rtamd_render instead. */
int rtamd_render_ppm(rtamd_scene *scene, const rtamd_camera *cam, const rtamd_render_params *p, uint8_t *pixels, int32_t *exact,
                     rtamd_stats *stats);

/* Page-locked host memory for the output planes of rtamd_render: the download then goes straight into the caller's
 * buffer at the link's rate (any page-locked buffer does: hipHostMalloc, hipHostRegister); pageable buffers work too
 * and cost an extra host copy.  Stands where the reference allocates pixel_data (src/flyscene.cpp:399-403). */
int rtamd_host_alloc(void **ptr, uint64_t bytes);
int rtamd_host_free(void *ptr);

/* Same work, enqueued on `hip_stream` (a hipStream_t; NULL = default stream) without any host
 * synchronisation; outputs are DEVICE pointers on the scene's device (face_id, t_hit may be NULL;
 * t_hit = local_rows*width floats, primary hit distance, 0 on a miss).  rtamd_render_stats()
 * waits for the frames enqueued so far and returns the counters of the most recent call.
 * Frames in flight: the scene keeps one workspace per stream it is rendered on (thirty-two at most;
 * one stream more takes over the least recently used one, after the host has waited for its last
 * frame), so frames enqueued on different streams may run at the same time -- the short, latency-
 * bound launches of one frame then fill the gaps of the other.  The GPU runs four kernels at a time
 * however many streams feed it: FOUR streams, frame k on stream k mod 4, reach the full rate (eight or sixteen
 * are no faster and hold that many workspaces; five or six are slower).  Streams overlap at all only while each
 * has a hardware queue of its own; the HIP runtime gives a process four unless GPU_MAX_HW_QUEUES says otherwise
 * before its first call.  A frame is enqueued on hip_stream and on no other stream. */
int rtamd_render_device(rtamd_scene *scene, const rtamd_camera *cam, const rtamd_render_params *p, float *rgb_dev,
                        int32_t *face_id_dev, float *t_hit_dev, void *hip_stream);
int rtamd_render_stats(rtamd_scene *scene, rtamd_stats *stats);

/* rtamd_render_device for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same arguments, same results, but the
 * work buffers of the frame (hit lists, ray queues, counters, screen bins) live in one static array of
 * RTAMD_NOALLOC_WORKSPACE_BYTES that is part of the library's code object -- every GPU the library runs on holds
 * that much until the process ends -- so the call allocates, frees and creates nothing: no hipMalloc, no events,
 * no internal streams (everything is enqueued on hip_stream; RTAMD_FLAG_TIME_KERNELS is ignored).  The array's
 * address is looked up when a scene is created on the device.
 * Limit: the buffers take, per pixel of the local image rounded up to 8x8 tiles and with n = lights *
 * samples_per_light, 40 + n bytes without reflections, 104 + 2 n with one bounce, 184 + 2 n beyond (640x480 with 13
 * shadow rays per hit fits at any depth).  An unpartitioned frame (num_parts == 1) that does not fit is rendered in
 * bands of rows, one after the other through the array (as under rtamd_scene_set_workspace_limit; 1920x1080 with one
 * bounce goes in five bands); a frame that does not fit even in as many bands as its passes allow
 * (bands * ssaa^2 * (max_reflections + 1) <= 64), or a partitioned one that does not fit, is refused with
 * RTAMD_ERR_LIMIT.  When the screen bins of the primary rays do not fit beside the rest, those rays walk the
 * BVH instead (same image).
 * All calls on one GPU share that GPU's array, whatever the scene: frames enqueued through this entry point must
 * not overlap on a device (enqueue them on one stream, or order the streams); two that do race. */
#define RTAMD_NOALLOC_WORKSPACE_BYTES (64u << 20)
int rtamd_render_device_noalloc(rtamd_scene *scene, const rtamd_camera *cam, const rtamd_render_params *p, float *rgb_dev,
                                int32_t *face_id_dev, float *t_hit_dev, void *hip_stream);

/* The 8-bit image writePPMImage stores (utils/ppmIO.hpp:188-209): bytes_dev[i] = min(255, (int)(255 *
 * rgb_dev[i])) for `count` channel values, on the device and on hip_stream.  With several GPUs the ranks
 * can exchange these bytes instead of the floats: a quarter of the traffic, the same result.ppm.
 * rgb_dev 16-byte aligned, bytes_dev 4-byte aligned. */
int rtamd_quantize_ppm(const float *rgb_dev, uint8_t *bytes_dev, uint64_t count, void *hip_stream);

/* Interleave the packed rows of `num_parts` parts (as gathered on one device, part-major) back
 * into image order.  src_dev = concatenation over parts of local_rows(part)*width*channels
 * elements of 4 bytes; dst_dev = height*width*channels.  Enqueued on hip_stream.  (Rows of 8-bit
 * pixels: pass width*3/4 as width with channels = 1 when width*3 is a multiple of 4.) */
int rtamd_assemble_rows(const rtamd_render_params *p, const void *src_dev, void *dst_dev, int32_t channels,
                        void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
