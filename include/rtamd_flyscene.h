/*
 * rtamd_flyscene.h -- flat C view of the GL-free Flyscene host (raytracer_amd/csrc/flyscene_host.hpp).
 *
 * The C++ class rtamd::Flyscene keeps the reference's entry points (src/flyscene.hpp:45-190:
 * initialize, getCamera, addLight, raytraceScene) without OpenGL, GLFW or Tucano; these functions
 * expose the same object to C, Python (ctypes) or any other FFI.  Same conventions as rtamd.h:
 * 0 = success, message via rtamd_last_error().
 */
#ifndef RTAMD_FLYSCENE_H
#define RTAMD_FLYSCENE_H

#include "rtamd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtfly rtfly;

int rtfly_create(rtfly **out);
int rtfly_destroy(rtfly *f);

/* Scene inputs; call before rtfly_initialize.  The reference hard-codes all of them
 * (src/flyscene.cpp:8-16,39; src/flyscene.hpp:22-41). */
int rtfly_set_model(rtfly *f, const char *obj_path);     /* default "resources/models/bunny.obj"; a name ending
                                                            in ".ply" is read as PLY (ascii or binary
                                                            little-endian, triangles), which the reference
                                                            only previews (utils/plyimporter.hpp:213-284) */
/* Mesh given as arrays instead of an OBJ file (file coordinates; normals may be NULL; one material
 * id per face, -1 = none).  Same derived quantities as the OBJ path. */
int rtfly_set_model_arrays(rtfly *f, const float *vertices, const float *normals, int32_t num_vertices,
                           const int32_t *faces, const int32_t *face_material, int32_t num_faces,
                           const rtamd_material *materials, int32_t num_materials);
/* Textures for a mesh given as arrays (after rtfly_set_model_arrays): what the OBJ path derives from `vt`,
 * `map_kd` and the PPM files (utils/objimporter.hpp:51-79; utils/mtlIO.hpp:122-125; src/flyscene.cpp:52-71).
 * face_texcoords = num_faces*6; material_textures = num_materials indices into textures, -1 = none. */
int rtfly_set_texture_arrays(rtfly *f, const float *face_texcoords, const int32_t *material_textures,
                             const rtamd_texture *textures, int32_t num_textures);
int rtfly_clear_lights(rtfly *f);                        /* drop the default light (-1,1,1)         */
int rtfly_add_light_at(rtfly *f, const float *position, const float *color);
int rtfly_set_shadow_sampling(rtfly *f, int32_t hard_shadow, int32_t soft_points, float soft_radius);
int rtfly_add_sphere(rtfly *f, const float *center, float radius);
int rtfly_set_normalize_model(rtfly *f, int32_t on);     /* mesh.normalizeModelMatrix(), default on */
int rtfly_set_output(rtfly *f, const char *ppm_path);    /* default "result.ppm"; "" = no file      */
int rtfly_set_build_params(rtfly *f, const rtamd_build_params *b);

/* Flyscene::initialize(width, height), src/flyscene.cpp:18-163 without the GL parts: load, flatten
 * (precomputeData), light samples (precomputeLights), BVH + upload (rtamd_scene_create).
 * With upload = 0 the device is not touched (host-side arrays only). */
int rtfly_initialize(rtfly *f, int32_t width, int32_t height, int32_t upload);

/* Flyscene::addLight, src/flyscene.hpp:76: a white light at the camera centre, appended to the light list
 * and nothing else.  Like the reference -- which computes the shadow sample points once, in initialize
 * (src/flyscene.cpp:117) -- it gives the new light NO sample points: the light is shaded (:505) but casts no
 * shadow rays (:880).  Before rtfly_initialize the light simply joins the list and is sampled with the rest. */
int rtfly_add_light(rtfly *f);

/* Camera (Tucano::Flycamera subset). */
int rtfly_camera_reset(rtfly *f);
int rtfly_camera_translate(rtfly *f, float dx, float dy, float dz);  /* utils/flycamera.hpp:196-202 */
int rtfly_camera_rotate(rtfly *f, float angle_x, float angle_y);     /* adds to rotation_X/Y_axis   */
int rtfly_camera_update_view(rtfly *f);                              /* updateViewMatrix, :166-191  */
int rtfly_get_camera(rtfly *f, rtamd_camera *cam);                   /* what raytraceScene reads    */

/* The run-time form of the reference's macros; edit in place before rendering. */
rtamd_render_params *rtfly_params(rtfly *f);

/* Flyscene::raytraceScene, src/flyscene.cpp:381-459: render the current view and write the PPM. */
int rtfly_raytrace_scene(rtfly *f);
/* Same frame into caller buffers (rgb: H*W*3, face_id: H*W or NULL), no file. */
int rtfly_raytrace_to(rtfly *f, float *rgb, int32_t *face_id, rtamd_stats *stats);

/* The uploaded scene, for the device-pointer entry points of rtamd.h (NULL before initialize). */
rtamd_scene *rtfly_scene(rtfly *f);

/* Flattened arrays (struct PrecomputedData, src/defs.h:7-20, and friends) for inspection.
 * name: "raw_vertices" "raw_normals" "vertices" "normals" "face_normals" "face_origin_distance"
 * "light_samples" "materials" "model_matrix" (float) or "face_vertex_ids" "face_material_ids"
 * (int32).  count = number of 4-byte elements. */
int rtfly_get_array(rtfly *f, const char *name, const void **data, int64_t *count);

/* ImageImporter::writePPMImage, utils/ppmIO.hpp:188-209. */
int rtfly_write_ppm(const char *path, const float *rgb, int32_t width, int32_t height);
/* Same quantisation (utils/ppmIO.hpp:203), written as binary P6.  rtfly_set_output with a name ending in
 * ".p6.ppm" or ".pnm" makes rtfly_raytrace_scene write this form. */
int rtfly_write_ppm_binary(const char *path, const float *rgb, int32_t width, int32_t height);

#ifdef __cplusplus
}
#endif
#endif
