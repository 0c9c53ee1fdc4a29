// GL-free host mirror of the reference's Flyscene (src/flyscene.hpp:45-190).
//
// Same entry points -- initialize(w, h), getCamera(), addLight(), raytraceScene() -- with the
// OpenGL preview, GLFW and Tucano removed.  Everything below raytraceScene() runs on the GPU through
// the C ABI of include/rtamd.h; this class only prepares what crosses that boundary:
// OBJ/MTL reading (utils/objimporter.hpp, utils/mtlIO.hpp), the unit-cube model matrix
// (model.hpp:169-173), Flyscene::precomputeData / precomputeLights (src/flyscene.cpp:165-250), the
// camera quantities tracePixels reads, and the PPM writer (utils/ppmIO.hpp:188-209).
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rtamd.h"

namespace rtamd {

struct NamedMaterial {
    std::string name;
    rtamd_material m;
    std::string diffuse_texture; // Mtl::diffuse_tex_filename (map_kd), empty = none
};

// texturedatas[filename] of the reference, src/flyscene.cpp:52-71
struct HostTexture {
    std::string file;
    int32_t width = 0, height = 0;
    std::vector<float> rgb;
};

// What Tucano::Mesh holds after MeshImporter::loadObjFile.
struct HostMesh {
    std::vector<float> vertices;      // 3 per vertex, file coordinates
    std::vector<float> normals;       // 3 per vertex
    std::vector<int32_t> face_vertex; // 3 per face, in file order
    std::vector<float> face_normal;   // 3 per face, Face::normal
    std::vector<int32_t> face_material;
    std::vector<float> face_texcoord; // 6 per face (Face::texcoord), empty when the mesh has no texture coordinates
    std::vector<NamedMaterial> materials;
    std::vector<HostTexture> textures;
    std::vector<int32_t> material_texture; // per material: index into textures, -1 none
    int32_t num_vertices() const { return (int32_t)(vertices.size() / 3); }
    int32_t num_faces() const { return (int32_t)(face_vertex.size() / 3); }
};

bool load_obj(const std::string &path, HostMesh &mesh, std::string &error);
// Binary or ASCII PLY with float x y z [nx ny nz] (other vertex properties skipped) and triangle faces.  The
// reference only previews PLY files (utils/plyimporter.hpp:213-284); here they feed the same pipeline as an
// OBJ.  Materials: `<name>.mtl` beside the file if present (its first material for every face), else none.
bool load_ply(const std::string &path, HostMesh &mesh, std::string &error);
bool load_mtl(const std::string &path, std::vector<NamedMaterial> &materials);
bool load_ppm_texture(const std::string &path, HostTexture &tex, std::string &error); // utils/ppmIO.hpp:36-86
bool load_material_textures(HostMesh &mesh, std::string &error);                       // src/flyscene.cpp:52-71
void finish_mesh(HostMesh &mesh, bool have_normals);
bool write_ppm(const std::string &path, const float *rgb, int width, int height);
// The same files (P3 text, or P6 with binary = true) from pixels already quantised to the file's 8 bits.
bool write_ppm_bytes(const std::string &path, const unsigned char *pixels, int width, int height, bool binary);
// Same quantisation as write_ppm, written as binary P6 (a tenth of the bytes, no text formatting).
bool write_ppm_binary(const std::string &path, const float *rgb, int width, int height);

// Subset of Tucano::Flycamera / Tucano::Camera that the traced path touches.
class Flycamera {
public:
    Flycamera() { reset(); }
    void reset();                                                            // utils/flycamera.hpp:76-86
    void setPerspectiveMatrix(float fovy, float aspect, float, float);       // camera.hpp:441-452 (only fovy, aspect kept)
    void setViewport(float w, float h);                                      // camera.hpp:327-330
    void translate(float dx, float dy, float dz);                            // utils/flycamera.hpp:196-202
    void rotate(float angle_x, float angle_y);                               // adds to rotation_X_axis / rotation_Y_axis
    void updateViewMatrix();                                                 // utils/flycamera.hpp:166-191
    void reComputeViewMatrix();                                              // camera.hpp:179-181
    std::array<float, 3> getCenter() const;                                  // camera.hpp:118-121
    std::array<int, 2> getViewportSize() const { return {(int)viewport_[2], (int)viewport_[3]}; }
    void fill(rtamd_camera &c) const;

private:
    float view_[12];     // row-major 3x4
    float inv_view_[12]; // inverseViewMatrix
    float viewport_[4];
    float fovy_ = 60.0f, aspect_ = 1.0f;
    float speed_ = 0.05f;
    float translation_[3];
    float rot_x_ = 0.0f, rot_y_ = 0.0f;
};

struct Light {
    float position[3];
    float color[3];
};

class Flyscene {
public:
    // --- the reference's public entry points
    int initialize(int width, int height, bool upload = true); // src/flyscene.cpp:18-163
    Flycamera *getCamera() { return &flycamera; }              // src/flyscene.hpp:71
    int addLight();                                            // src/flyscene.hpp:76
    int raytraceScene();                                       // src/flyscene.cpp:381-459
    ~Flyscene();

    // --- inputs the reference hard-codes
    std::string model_path = "resources/models/bunny.obj"; // src/flyscene.cpp:39
    std::string output_path = "result.ppm";                // src/flyscene.cpp:445
    bool hard_shadow = true;                               // HARDSHADOW
    int soft_points = 12;                                  // MAXSOFTSHADOWPOINTS
    float soft_radius = 0.3f;                              // SOFTSHADOWRADIUS
    bool normalize_model = true;                           // src/flyscene.cpp:49
    std::vector<Light> lights;                             // loadLightsAndSpheres, src/flyscene.cpp:8-16
    std::vector<rtamd_sphere> spheres;
    rtamd_render_params params{};
    rtamd_build_params build{};
    bool lights_defaulted = false;
    bool mesh_from_arrays = false;
    HostMesh mesh;

    // --- struct PrecomputedData, src/defs.h:7-20
    std::vector<float> P_vertices, P_normals, P_face_distance;
    std::vector<float> light_samples; // PrecomputedData::lights: sample points of the first sampled_lights entries of `lights`
    size_t sampled_lights = 0;        // lights.size() when precomputeLights last ran (the reference runs it once, src/flyscene.cpp:117)
    float model_matrix[12];
    std::vector<rtamd_material> plain_materials;

    int render(float *rgb, int32_t *face_id, rtamd_stats *stats);
    rtamd_scene *scene() { return scene_; }
    void precomputeData();   // src/flyscene.cpp:165-219
    void precomputeLights(); // src/flyscene.cpp:221-250
    int upload();

private:
    void loadLightsAndSpheres(); // src/flyscene.cpp:8-16
    Flycamera flycamera;
    rtamd_scene *scene_ = nullptr;
    bool initialized_ = false;
    void *pixel_bytes_ = nullptr; // page-locked: the 8-bit frame raytraceScene downloads
    size_t pixel_bytes_size_ = 0;
};

} // namespace rtamd
