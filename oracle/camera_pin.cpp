// Pins the camera leg of the path against the reference's own Tucano::Camera.  TEST INFRASTRUCTURE ONLY.
// Built by oracle/Makefile into oracle/_ref/camera_pin from the reference's headers where they lie
// (dependencies/tucano, dependencies/eigen, dependencies/glew/include with GLEW's documented GLEW_NO_GLU
// switch); no header, library or symbol is replaced.  Never shipped, never linked into the product.
//
// Reference code that runs here: Camera::setPerspectiveMatrix, setViewport, resetViewMatrix, viewMatrix,
// reComputeViewMatrix, getCenter, getPerspectiveScale, screenToWorld (camera.hpp:91-94,118-121,158-181,222-225,
// 271-274,327-330,441-452) -- what Flyscene::initialize (src/flyscene.cpp:35-36) and tracePixels (:806,820)
// call.  Tucano::Flycamera itself does not link without OpenGL (its CoordinateAxes member constructs a GL
// mesh), so its reset / updateViewMatrix / translate / rotation angles (utils/flycamera.hpp:76-86,166-202)
// are driven here step by step on the Camera's own view matrix with the same Eigen calls
// (AngleAxisf products, Affine3f::rotate / translate): that composition is a restatement, the Eigen
// arithmetic under it and everything downstream is the reference's.
//
// usage: camera_pin OUT W H YSTEP [t dx dy dz | r ax ay]...
// writes to OUT (binary, little endian): float origin[3], float inv_view[12] (row-major 3x4), then for every
// y in 0, YSTEP, 2*YSTEP, ... < H and every x in [0, W): float world[3] = screenToWorld((x, y)).
#include <tucano/camera.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: camera_pin OUT W H YSTEP [t dx dy dz | r ax ay]...\n");
        return 2;
    }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), ystep = std::atoi(argv[4]);
    if (W <= 0 || H <= 0 || ystep <= 0) return 2;

    Tucano::Camera cam;
    // Flyscene::initialize, src/flyscene.cpp:35-36
    cam.setPerspectiveMatrix(60.0, (float)W / (float)H, 0.1f, 100.0f);
    cam.setViewport(Eigen::Vector2f((float)W, (float)H));

    // Flycamera state, utils/flycamera.hpp:47-66,76-86
    const float speed = 0.05;
    Eigen::Vector3f translation_vector = Eigen::Vector3f::Zero();
    Eigen::Matrix3f rotation_matrix = Eigen::Matrix3f::Identity();
    const Eigen::Vector3f default_translation(0.0, 0.0, -2.0);
    const Eigen::Matrix3f default_rotation = Eigen::Matrix3f::Identity();
    float rotation_X_axis = 0.0, rotation_Y_axis = 0.0;

    for (int a = 5; a < argc;) {
        if (!std::strcmp(argv[a], "t") && a + 3 < argc) { // Flycamera::translate, :196-202
            const float dx = (float)std::atof(argv[a + 1]), dy = (float)std::atof(argv[a + 2]), dz = (float)std::atof(argv[a + 3]);
            Eigen::AngleAxisf yaw = Eigen::AngleAxisf(rotation_Y_axis, Eigen::Vector3f::UnitY());
            translation_vector += yaw * Eigen::Vector3f(-dx, -dy, dz) * speed;
            a += 4;
        } else if (!std::strcmp(argv[a], "r") && a + 2 < argc) { // the angle bookkeeping of Flycamera::rotate, :291-292
            rotation_X_axis += (float)std::atof(argv[a + 1]);
            rotation_Y_axis += (float)std::atof(argv[a + 2]);
            a += 3;
        } else {
            std::fprintf(stderr, "camera_pin: bad operation '%s'\n", argv[a]);
            return 2;
        }
    }
    // Flycamera::updateViewMatrix, :166-191
    cam.resetViewMatrix();
    Eigen::Vector3f rotX = Eigen::AngleAxisf(rotation_Y_axis, Eigen::Vector3f::UnitY()) * Eigen::Vector3f::UnitX();
    rotX.normalize();
    Eigen::Vector3f rotZ = Eigen::AngleAxisf(rotation_Y_axis, Eigen::Vector3f::UnitY()) * Eigen::Vector3f::UnitZ();
    rotZ = Eigen::AngleAxisf(rotation_X_axis, rotX) * rotZ;
    rotZ.normalize();
    Eigen::Vector3f rotY = Eigen::AngleAxisf(rotation_X_axis, rotX) * Eigen::Vector3f::UnitY();
    rotY.normalize();
    rotation_matrix.row(0) = rotX;
    rotation_matrix.row(1) = rotY;
    rotation_matrix.row(2) = rotZ;
    cam.viewMatrix()->rotate(default_rotation);
    cam.viewMatrix()->rotate(rotation_matrix);
    cam.viewMatrix()->translate(default_translation);
    cam.viewMatrix()->translate(translation_vector);

    cam.reComputeViewMatrix(); // tracePixels, src/flyscene.cpp:806
    const Eigen::Vector3f origin = cam.getCenter(); // raytraceScene, src/flyscene.cpp:413
    const Eigen::Affine3f inv = cam.getViewMatrix().inverse(); // what reComputeViewMatrix stored (camera.hpp:180)

    std::FILE *out = std::fopen(argv[1], "wb");
    if (!out) return 1;
    float head[15];
    for (int k = 0; k < 3; ++k) head[k] = origin[k];
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) head[3 + 4 * r + k] = inv.linear()(r, k);
        head[3 + 4 * r + 3] = inv.translation()(r);
    }
    std::fwrite(head, sizeof(float), 15, out);
    std::vector<float> row((size_t)W * 3);
    for (int y = 0; y < H; y += ystep) {
        for (int x = 0; x < W; ++x) {
            const Eigen::Vector3f p = cam.screenToWorld(Eigen::Vector2f((float)x, (float)y)); // src/flyscene.cpp:820
            row[3 * (size_t)x] = p[0], row[3 * (size_t)x + 1] = p[1], row[3 * (size_t)x + 2] = p[2];
        }
        std::fwrite(row.data(), sizeof(float), row.size(), out);
    }
    std::fclose(out);
    return 0;
}
