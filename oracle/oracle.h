/*
 * CPU oracle for the Flyscene ray-tracing pass.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference renderer's algorithm, used as the checker
 * in tests, in the smoke test and as the timed CPU baseline of the benchmark.  No
 * product code may include, link or call it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer
 * vectors, and it cannot be compiled in this environment (merge residue in
 * src/flyscene.cpp, hard dependency on OpenGL/GLEW/GLFW even for its BVH nodes).
 * The only reference code this restatement is checked against mechanically is the
 * vendored Eigen 3.3.7 (association order of dot/norm/cross/affine products, see
 * eigen_pin.cpp).  Everything else follows the reference source by reading it.
 *
 * Canonical arithmetic: IEEE-754 binary32, source evaluation order, no FMA
 * contraction, correctly rounded division and sqrt; the few double intermediates
 * of the reference (camera.hpp:162-168) are kept.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* Material as read by mtlIO.hpp:49-140 with the defaults of mtl.hpp:22-31. */
typedef struct {
    float ka[3];
    float kd[3];
    float ks[3];
    float shininess;
} orc_material;

/* What tracePixels/screenToWorld read from the camera (camera.hpp:84,158-181). */
typedef struct {
    float origin[3];    /* flycamera.getCenter(), camera.hpp:118-121            */
    float inv_view[12]; /* inverseViewMatrix, row-major 3x4 = [linear | transl] */
    float viewport[4];  /* (0, 0, W, H), camera.hpp:327-330                     */
    float aspect_ratio; /* fixed at initialize(): (float)w/(float)h             */
    float fovy;         /* degrees; 60 in flyscene.cpp:35                       */
} orc_camera;

/* Compile-time macros of flyscene.hpp:22-41 and literals of flyscene.cpp as run-time fields. */
typedef struct {
    int32_t width, height;        /* viewport size, flyscene.cpp:394-396        */
    int32_t ssaa_level;           /* SSAALEVEL, flyscene.hpp:36                 */
    int32_t max_reflections;      /* MAXREFLECTIONS, flyscene.hpp:40            */
    float reflect_threshold;      /* 0.25, flyscene.cpp:631                     */
    float bounce_offset;          /* 0.01, flyscene.cpp:636                     */
    float t_epsilon;              /* 1e-5, flyscene.cpp:707,760                 */
    float background[3];          /* 0.4, flyscene.cpp:653-658                  */
    float background_deep[3];     /* 0.0, flyscene.cpp:661-665                  */
    int32_t num_threads;          /* 0 = hardware_concurrency, flyscene.cpp:415 */
    int32_t row_offset, row_step; /* render only rows y = row_offset + k*row_step (bounded timing samples); 0,1 = all */
    int32_t reject_nan;           /* 0 = as the reference: a NaN plane distance passes both range tests and is accepted
                                     (flyscene.cpp:760), and a 0 / 0 in a box test (a ray IN the plane of a box side,
                                     boundingBox.h:258-271) turns away or takes the box depending on the side; 1 = such
                                     faces are skipped and such a slab counts as passed (no decision is left to a 0 / 0) */
    int32_t all_faces;            /* 1 = no box turns a ray away: every leaf is visited, in the reference's order, and the faces
                                     are accepted by the face test alone (flyscene.cpp:752-789).  The reference's own box test
                                     (boundingBox.h:257-286) rounds, and a ray through a corner of a leaf's box may be turned
                                     away from a face that the face test would accept; which rays depends on the boxes the
                                     hierarchy happens to have.  The HIP path has other boxes and defines the outcome by the
                                     face test alone: it is compared with this mode where the two differ. */
} orc_params;

typedef struct {
    uint64_t rays_primary;   /* intersects() calls from tracePixels            */
    uint64_t rays_shadow;    /* intersects() calls from getLightIntensity      */
    uint64_t rays_bounce;    /* intersects() calls from recursive traceRay     */
    uint64_t box_tests;      /* boundingBox::boxIntersection calls             */
    uint64_t leaves_visited; /* leaves appended by intersectingBoxes           */
    uint64_t tri_plane;      /* plane-distance evaluations, flyscene.cpp:758   */
    uint64_t tri_edge;       /* faces that reached the edge tests, :771-780    */
    uint64_t nan_accepts;    /* hits accepted with NaN t (degenerate faces)    */
    double seconds;          /* spawn -> join wall time, flyscene.cpp:422-442  */
} orc_counters;

/* S1+S2: objimporter.hpp:117-351, mtlIO.hpp:49-140, mesh.hpp:448-482.  NULL on open failure. */
orc_scene *orc_scene_load_obj(const char *path);
/* Same mesh semantics from arrays (raw, un-normalised coordinates; normals may be NULL ->
 * recomputed as objimporter.hpp:81-106; face_group_material gives one material id per face). */
orc_scene *orc_scene_from_arrays(const float *vertices, const float *normals, int32_t nv,
                                 const int32_t *faces, const int32_t *face_material, int32_t nf,
                                 const orc_material *materials, int32_t nmat);
void orc_scene_free(orc_scene *s);

/* Texture path (shadeOffFace :516-542).  orc_scene_load_obj reads `vt`, `map_kd` and the PPM files itself;
 * these serve scenes built from arrays.  texcoords = nf*6 (u,v per corner, Face::texcoord) or NULL;
 * rgb = width*height*3 in the row order of texturedatas (file rows flipped, ppmIO.hpp:68-77). */
void orc_scene_set_texcoords(orc_scene *s, const float *texcoords);
int32_t orc_scene_add_texture(orc_scene *s, int32_t width, int32_t height, const float *rgb);
void orc_scene_set_material_texture(orc_scene *s, int32_t material, int32_t texture);
const float *orc_face_texcoords(const orc_scene *s);          /* nf*6 or NULL                      */
int32_t orc_num_textures(const orc_scene *s);
const float *orc_texture_data(const orc_scene *s, int32_t i, int32_t *width, int32_t *height);
const int32_t *orc_material_textures(const orc_scene *s);     /* nmat or NULL                      */

/* S3+S4: mesh.normalizeModelMatrix() + Flyscene::precomputeData, flyscene.cpp:165-219. */
void orc_scene_precompute(orc_scene *s, int32_t normalize_model);
/* lights: pos_color = L x 6 floats.  S5: Flyscene::precomputeLights, flyscene.cpp:221-250. */
void orc_scene_set_lights(orc_scene *s, const float *pos_color, int32_t nlights, int32_t hard_shadow,
                          int32_t soft_points, float soft_radius);
/* Flyscene::addLight, flyscene.hpp:76: appends one light (pos_color = 6 floats) WITHOUT shadow sample points --
 * precomputeLights ran once, in initialize (flyscene.cpp:117).  shadeOffFace loops over it (:505),
 * getLightIntensity (:880) does not. */
void orc_scene_add_light(orc_scene *s, const float *pos_color);
/* spheres = n x 4 floats (center, radius), defs.h:22-25. */
void orc_scene_set_spheres(orc_scene *s, const float *spheres, int32_t n);
/* B1+B2: boundingVectors (flyscene.cpp:841-868) + boundingBox::splitBox (boundingBox.h:69-194). */
void orc_scene_build_bvh(orc_scene *s);

/* Accessors (pointers stay owned by the scene). */
int32_t orc_num_vertices(const orc_scene *s);
int32_t orc_num_faces(const orc_scene *s);
int32_t orc_num_materials(const orc_scene *s);
int32_t orc_num_lights(const orc_scene *s);          /* lights with shadow sample points            */
int32_t orc_num_shading_lights(const orc_scene *s);  /* all lights, those of orc_scene_add_light too */
int32_t orc_samples_per_light(const orc_scene *s);
const float *orc_raw_vertices(const orc_scene *s);          /* nv*3, file coordinates           */
const float *orc_raw_normals(const orc_scene *s);           /* nv*3, before normalisation       */
const float *orc_vertices(const orc_scene *s);              /* nv*3, PrecomputedData::vertices  */
const float *orc_normals(const orc_scene *s);               /* nv*3                              */
const int32_t *orc_face_vertex_ids(const orc_scene *s);     /* nf*3                              */
const float *orc_face_normals(const orc_scene *s);          /* nf*3                              */
const float *orc_face_origin_distance(const orc_scene *s);  /* nf                                */
const int32_t *orc_face_material_ids(const orc_scene *s);   /* nf                                */
const orc_material *orc_materials(const orc_scene *s);      /* nmat                              */
const float *orc_light_samples(const orc_scene *s);         /* L*S*3                             */
const float *orc_model_matrix(const orc_scene *s);          /* 12 floats row-major 3x4           */
int32_t orc_bvh_node_count(const orc_scene *s);
int32_t orc_bvh_leaf_count(const orc_scene *s);
int32_t orc_bvh_depth(const orc_scene *s);
/* Position of every face in the leaf sequence of a preorder (lower, upper) walk: the order in
 * which triangleIntersection would meet the faces if every leaf were hit.  out = nf ints. */
void orc_bvh_face_order(const orc_scene *s, int32_t *out);
/* Preorder dump: per node 6 floats (vmin,vmax) and 2 ints (is_leaf, nfaces). */
void orc_bvh_dump(const orc_scene *s, float *boxes, int32_t *info);

void orc_camera_default(orc_camera *c, int32_t width, int32_t height);
void orc_params_default(orc_params *p, int32_t width, int32_t height);

/* R1..R10.  Outputs may be NULL.  rgb = H*W*3 row-major, row 0 = top; face_id: -2 miss, -1 sphere,
 * else face; t_hit: primary t (0 on miss); light = light intensity at the primary hit (0 on miss).
 * With ssaa_level > 1 the per-pixel planes hold the LAST sub-sample.  nan_mask: 1 where any ray of the
 * pixel's ray tree met a NaN plane distance (ray inside a face plane: 0/0, or a zero-area face); the
 * reference's result for such a pixel depends on its BVH visiting order. */
int32_t orc_render(orc_scene *s, const orc_camera *cam, const orc_params *p, float *rgb, int32_t *face_id,
                   float *t_hit, float *light, int32_t *nan_mask, orc_counters *counters);

/* Single-ray probes for unit tests. */
int32_t orc_intersects(orc_scene *s, const float *origin, const float *direction, const orc_params *p,
                       int32_t *face_id, float *t_hit, float *hitpoint, float *reflection);
void orc_screen_ray(const orc_camera *cam, float px, float py, float *direction);
/* Camera::screenToWorld (camera.hpp:158-176) of every pixel corner (x, y), x in [0, width), y = 0, ystep, 2*ystep, ...
 * < height: the world point BEFORE tracePixels subtracts the origin and normalises.  out = rows*width*3 floats. */
void orc_screen_points(const orc_camera *cam, int32_t width, int32_t height, int32_t ystep, float *out);

/* O1: ppmIO.hpp:188-209. */
int32_t orc_write_ppm(const char *path, const float *rgb, int32_t width, int32_t height);

#ifdef __cplusplus
}
#endif
#endif
