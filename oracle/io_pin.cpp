// Pins the image writer of the path against the reference's own ImageImporter::writePPMImage.
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/io_pin from the reference's headers where they
// lie; no header, library or symbol is replaced (`-include fstream`: utils/ppmIO.hpp uses std::ofstream without
// including its header; GLEW_NO_GLU is GLEW's own switch).  Never shipped, never linked into the product.
//
// Reference code that runs here: writePPMImage(string, const vector<vector<Eigen::Vector3f>>&), utils/ppmIO.hpp:188-209,
// the call Flyscene::raytraceScene ends with (src/flyscene.cpp:445).
// NOT pinnable the same way: loadPPMImageData (utils/ppmIO.hpp:36-86) calls Texture::create, and loadMTL
// (utils/mtlIO.hpp:49-140) constructs Tucano::Material::Mtl, which holds a Texture; both need glGenTextures,
// glDeleteTextures, glBindTexture, glTexImage1D/2D, glTexParameteri, glGetIntegerv, __glewTexImage2DMultisample and
// __glewTexImage3D at link time.  This file references neither.
//
// usage: io_pin IN OUT   IN = binary: int32 width, int32 height, float rgb[height][width][3]; OUT = the PPM file.
#include <tucano/utils/ppmIO.hpp>

#include <cstdint>
#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    if (argc != 3) {
        std::fprintf(stderr, "usage: io_pin IN OUT\n");
        return 2;
    }
    std::FILE *in = std::fopen(argv[1], "rb");
    if (!in) return 1;
    int32_t size[2];
    if (std::fread(size, sizeof(int32_t), 2, in) != 2 || size[0] <= 0 || size[1] <= 0) return 1;
    std::vector<std::vector<Eigen::Vector3f>> pixel_data; // src/flyscene.cpp:399-403
    pixel_data.resize((size_t)size[1]);
    std::vector<float> row((size_t)size[0] * 3);
    for (int j = 0; j < size[1]; ++j) {
        if (std::fread(row.data(), sizeof(float), row.size(), in) != row.size()) return 1;
        pixel_data[(size_t)j].resize((size_t)size[0]);
        for (int i = 0; i < size[0]; ++i) pixel_data[(size_t)j][(size_t)i] = Eigen::Vector3f(row[3 * (size_t)i], row[3 * (size_t)i + 1], row[3 * (size_t)i + 2]);
    }
    std::fclose(in);
    Tucano::ImageImporter::writePPMImage(argv[2], pixel_data);
    return 0;
}
