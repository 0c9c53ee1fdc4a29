#include "flyscene_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "../../include/rtamd_flyscene.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace rtamd {

namespace {

// Fixed-size float vector helpers in Eigen's evaluation order (3-term sums associate x + (y + z)).
struct F3 {
    float x, y, z;
};
inline F3 f3(const float *p) { return F3{p[0], p[1], p[2]}; }
inline F3 sub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(F3 a, F3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline F3 cross(F3 a, F3 b) { return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline F3 unit(F3 a) {
    const float z = dot(a, a);
    if (z > 0.0f) {
        const float s = std::sqrt(z);
        return F3{a.x / s, a.y / s, a.z / s};
    }
    return a;
}
inline void put(float *p, F3 a) { p[0] = a.x, p[1] = a.y, p[2] = a.z; }

void erase_crlf(std::string &s) {
    s.erase(std::remove(s.begin(), s.end(), '\n'), s.end());
    s.erase(std::remove(s.begin(), s.end(), '\r'), s.end());
}

std::string directory_of(const std::string &path) {
    const size_t cut = path.find_last_of("/\\");
    return cut == std::string::npos ? std::string() : path.substr(0, cut + 1);
}

rtamd_material default_material() { // materials/mtl.hpp:22-31
    rtamd_material m;
    for (int i = 0; i < 3; ++i) m.ka[i] = (float)0.3, m.kd[i] = (float)0.5, m.ks[i] = (float)1.0;
    m.shininess = 10;
    return m;
}

// Face::normal, mesh.hpp:475-477: unit cross product of the two unit edge vectors, file coordinates.
F3 raw_face_normal(const HostMesh &m, const int32_t *id) {
    const F3 p0 = f3(&m.vertices[3 * (size_t)id[0]]), p1 = f3(&m.vertices[3 * (size_t)id[1]]), p2 = f3(&m.vertices[3 * (size_t)id[2]]);
    const F3 e1 = unit(sub(p2, p0));
    const F3 e0 = unit(sub(p1, p0));
    return unit(cross(e0, e1));
}

} // namespace

// utils/mtlIO.hpp:49-140.  Records are split on single blanks, exactly like std::getline(ss, s, ' ').
bool load_mtl(const std::string &path, std::vector<NamedMaterial> &materials) {
    std::ifstream in(path.c_str());
    if (!in) {
        std::fprintf(stderr, "Cannot open %s\n", path.c_str());
        return false;
    }
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::vector<std::string> tok;
        size_t pos = 0;
        while (pos < line.size()) {
            const size_t sp = line.find(' ', pos);
            if (sp == std::string::npos) {
                tok.push_back(line.substr(pos));
                break;
            }
            tok.push_back(line.substr(pos, sp - pos));
            pos = sp + 1;
        }
        if (tok.empty()) continue;
        auto num = [&](size_t i) { return i < tok.size() ? (float)std::atof(tok[i].c_str()) : 0.0f; };
        const std::string &key = tok[0];
        if (key == "#") continue;
        if (key == "newmtl") {
            NamedMaterial nm;
            nm.m = default_material();
            nm.name = tok.size() > 1 ? tok[1] : std::string();
            erase_crlf(nm.name);
            materials.push_back(nm);
        } else if (!materials.empty()) {
            rtamd_material &m = materials.back().m;
            if (key == "Ns") {
                m.shininess = num(1);
            } else if (key == "Ka") {
                m.ka[0] = num(1), m.ka[1] = num(2), m.ka[2] = num(3);
            } else if (key == "Kd") {
                m.kd[0] = num(1), m.kd[1] = num(2), m.kd[2] = num(3);
            } else if (key == "Ks") {
                m.ks[0] = num(1), m.ks[1] = num(2), m.ks[2] = num(3);
            } else if (key == "map_kd") { // lower case only, :122-125: directory of the MTL file + the name as written
                materials.back().diffuse_texture = directory_of(path) + (tok.size() > 1 ? tok[1] : std::string());
            }
            // Ni, d and illum are parsed by the reference but not read by the traced path
        }
    }
    if (materials.empty()) { // :130-134
        NamedMaterial nm;
        nm.m = default_material();
        materials.push_back(nm);
    }
    return true;
}

// ImageImporter::loadPPMImageData, utils/ppmIO.hpp:36-86: header word, width, height, maximum, then every
// number divided by the maximum; the rows are handed back bottom row first (:68-77).  The reference goes on
// with empty data when the file is missing or short and then indexes it; here that is an error.
bool load_ppm_texture(const std::string &path, HostTexture &tex, std::string &error) {
    std::string fn = path;
    erase_crlf(fn); // :40-41
    tex = HostTexture();
    tex.file = fn;
    std::ifstream in(fn.c_str());
    if (!in) {
        error = "Cannot open " + fn;
        return false;
    }
    std::string header;
    int w = 0, h = 0;
    float max_value = 0.0f;
    in >> header;
    in >> w >> h;
    in >> max_value;
    if (!in || w <= 0 || h <= 0) {
        error = "not an ASCII PPM image: " + fn;
        return false;
    }
    std::vector<float> data;
    data.reserve((size_t)w * h * 3);
    float value;
    while (data.size() < (size_t)w * h * 3 && in >> value) data.push_back(value / max_value); // :57-60
    if (data.size() < (size_t)w * h * 3) {
        error = "texture " + fn + " holds fewer than width*height*3 values";
        return false;
    }
    tex.width = w, tex.height = h;
    tex.rgb.resize(data.size());
    for (int j = h - 1, r = 0; j >= 0; --j, ++r)
        std::memcpy(&tex.rgb[(size_t)r * w * 3], &data[(size_t)j * w * 3], (size_t)w * 3 * sizeof(float));
    return true;
}

// Flyscene::initialize, src/flyscene.cpp:52-71: one entry per distinct file name (std::map::insert keeps the first).
bool load_material_textures(HostMesh &mesh, std::string &error) {
    mesh.textures.clear();
    mesh.material_texture.assign(mesh.materials.size(), -1);
    for (size_t i = 0; i < mesh.materials.size(); ++i) {
        std::string fn = mesh.materials[i].diffuse_texture;
        if (fn.empty()) continue;
        erase_crlf(fn);
        for (size_t k = 0; k < mesh.textures.size(); ++k)
            if (mesh.textures[k].file == fn) mesh.material_texture[i] = (int32_t)k;
        if (mesh.material_texture[i] >= 0) continue;
        HostTexture t;
        if (!load_ppm_texture(fn, t, error)) return false;
        mesh.material_texture[i] = (int32_t)mesh.textures.size();
        mesh.textures.push_back(std::move(t));
    }
    return true;
}

// Vertex normals when the file gives none (or a different count): objimporter.hpp:81-106.
static void accumulate_vertex_normals(HostMesh &m) {
    std::fill(m.normals.begin(), m.normals.end(), 0.0f);
    const int32_t nf = m.num_faces();
    for (int32_t f = 0; f < nf; ++f) {
        const int32_t *id = &m.face_vertex[3 * (size_t)f];
        const F3 n = raw_face_normal(m, id);
        for (int k = 0; k < 3; ++k) {
            float *d = &m.normals[3 * (size_t)id[k]];
            d[0] += n.x, d[1] += n.y, d[2] += n.z;
        }
    }
    for (int32_t i = 0; i < m.num_vertices(); ++i) put(&m.normals[3 * (size_t)i], unit(f3(&m.normals[3 * (size_t)i])));
}

void finish_mesh(HostMesh &m, bool have_normals) {
    m.normals.resize(m.vertices.size(), 0.0f);
    if (!have_normals) accumulate_vertex_normals(m);
    const int32_t nf = m.num_faces();
    m.face_normal.assign(3 * (size_t)nf, 0.0f);
    for (int32_t f = 0; f < nf; ++f) put(&m.face_normal[3 * (size_t)f], raw_face_normal(m, &m.face_vertex[3 * (size_t)f]));
}

// utils/objimporter.hpp:117-351 + Mesh::createFaces (mesh.hpp:448-482).
//  * faces are numbered in the order of the `f` records (index groups per `usemtl` are appended in
//    file order and emitted group by group);
//  * every element of an `f` record contributes its vertex index -- no triangulation; faces are the
//    consecutive index triples of each group;
//  * the normal indices of `f` records are ignored: `vn` k belongs to `v` k when the counts agree,
//    otherwise vertex normals are recomputed from the faces;
//  * texture coordinates (objimporter.hpp:51-79,222-228,249-256,325-337; mesh.hpp:463-468) are kept per
//    VERTEX and per group: the j-th texture index of a group is written to the vertex named by the j-th
//    vertex index of that group (default (-1,-1), later records overwrite earlier ones), and the faces of
//    the k-th non-empty group read table k of all groups.
bool load_obj(const std::string &path, HostMesh &mesh, std::string &error) {
    std::ifstream in(path.c_str());
    if (!in) {
        error = "Cannot open " + path;
        return false;
    }
    mesh = HostMesh();
    const std::string dir = directory_of(path);
    std::vector<float> file_normals;
    std::vector<float> file_texcoords;               // 2 per `vt`
    std::vector<std::vector<int32_t>> group_tex(1);  // elementsTexIDs
    std::vector<std::vector<int32_t>> groups(1);
    std::vector<int32_t> group_material(1, -1);
    int32_t current_material = -1;
    std::string line;
    while (std::getline(in, line)) {
        if (line.compare(0, 6, "mtllib") == 0 && line.size() >= 7) {
            std::string fn = dir + line.substr(7);
            erase_crlf(fn);
            load_mtl(fn, mesh.materials);
        } else if (line.compare(0, 6, "usemtl") == 0 && line.size() >= 7) {
            if (!groups.back().empty()) {
                groups.emplace_back();
                group_tex.emplace_back();
                group_material.push_back(-1);
            }
            std::string name = line.substr(7);
            erase_crlf(name);
            for (size_t i = 0; i < mesh.materials.size(); ++i)
                if (mesh.materials[i].name == name) current_material = (int32_t)i; // sticky when not found
            group_material.back() = current_material;
        } else if (line.compare(0, 2, "v ") == 0) {
            const char *p = line.c_str() + 2;
            char *e;
            for (int k = 0; k < 3; ++k) {
                mesh.vertices.push_back(std::strtof(p, &e));
                p = e;
            }
        } else if (line.compare(0, 2, "vn") == 0) {
            const char *p = line.c_str() + std::min<size_t>(3, line.size());
            char *e;
            for (int k = 0; k < 3; ++k) {
                file_normals.push_back(std::strtof(p, &e));
                p = e;
            }
        } else if (line.compare(0, 2, "vt") == 0) {
            const char *p = line.c_str() + 2;
            char *e;
            for (int k = 0; k < 2; ++k) {
                file_texcoords.push_back(std::strtof(p, &e));
                p = e;
            }
        } else if (line.compare(0, 2, "f ") == 0) {
            std::istringstream els(line.substr(2));
            std::string el;
            while (els >> el) {
                groups.back().push_back((int32_t)std::strtol(el.c_str(), nullptr, 10) - 1);
                const size_t slash = el.find('/');
                if (slash != std::string::npos && slash + 1 < el.size() && el[slash + 1] != '/')
                    group_tex.back().push_back((int32_t)std::strtol(el.c_str() + slash + 1, nullptr, 10) - 1);
            }
        }
    }
    const int32_t nv = mesh.num_vertices();
    const int32_t nvt = (int32_t)(file_texcoords.size() / 2);
    std::vector<std::vector<float>> tables; // createTexCoordsArrays: one per-vertex table per group
    if (nvt > 0) {
        tables.assign(groups.size(), std::vector<float>(2 * (size_t)nv, -1.0f));
        for (size_t g = 0; g < groups.size(); ++g)
            for (size_t j = 0; j < group_tex[g].size() && j < groups[g].size(); ++j) {
                const int32_t vid = groups[g][j], tid = group_tex[g][j];
                if (vid < 0 || vid >= nv) continue; // reported below, with the faces
                if (tid < 0 || tid >= nvt) {
                    error = "face references texture coordinate " + std::to_string(tid + 1) + " of " + std::to_string(nvt) + " in " + path;
                    return false;
                }
                tables[g][2 * (size_t)vid] = file_texcoords[2 * (size_t)tid];
                tables[g][2 * (size_t)vid + 1] = file_texcoords[2 * (size_t)tid + 1];
            }
    }
    size_t nonempty = 0;
    for (size_t g = 0; g < groups.size(); ++g) {
        if (groups[g].empty()) continue; // only non-empty groups become index buffers, objimporter.hpp:315-322
        for (size_t i = 0; i + 2 < groups[g].size(); i += 3) {
            for (int k = 0; k < 3; ++k) {
                const int32_t id = groups[g][i + k];
                if (id < 0 || id >= nv) {
                    error = "face references vertex " + std::to_string(id + 1) + " of " + std::to_string(nv) + " in " + path;
                    return false;
                }
                mesh.face_vertex.push_back(id);
                if (nvt > 0) {
                    mesh.face_texcoord.push_back(tables[nonempty][2 * (size_t)id]);
                    mesh.face_texcoord.push_back(tables[nonempty][2 * (size_t)id + 1]);
                }
            }
            mesh.face_material.push_back(group_material[g]);
        }
        ++nonempty;
    }
    const bool paired = (int32_t)(file_normals.size() / 3) == nv && nv > 0;
    if (paired) mesh.normals = file_normals;
    finish_mesh(mesh, paired);
    return load_material_textures(mesh, error);
}

namespace {
struct PlyProp {
    std::string name;
    int bytes;      // size of the scalar, or of one list item
    int count_bytes; // > 0: a list, size of its count field
    char kind;      // 'f' float32, 'd' float64, 'i' signed, 'u' unsigned
};
bool ply_type(const std::string &t, int &bytes, char &kind) {
    if (t == "float" || t == "float32") return bytes = 4, kind = 'f', true;
    if (t == "double" || t == "float64") return bytes = 8, kind = 'd', true;
    if (t == "char" || t == "int8") return bytes = 1, kind = 'i', true;
    if (t == "uchar" || t == "uint8") return bytes = 1, kind = 'u', true;
    if (t == "short" || t == "int16") return bytes = 2, kind = 'i', true;
    if (t == "ushort" || t == "uint16") return bytes = 2, kind = 'u', true;
    if (t == "int" || t == "int32") return bytes = 4, kind = 'i', true;
    if (t == "uint" || t == "uint32") return bytes = 4, kind = 'u', true;
    return false;
}
double ply_scalar(const unsigned char *p, int bytes, char kind) {
    switch (kind) {
    case 'f': { float v; std::memcpy(&v, p, 4); return v; }
    case 'd': { double v; std::memcpy(&v, p, 8); return v; }
    case 'i':
        if (bytes == 1) return (signed char)p[0];
        if (bytes == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
        { int32_t v; std::memcpy(&v, p, 4); return v; }
    default:
        if (bytes == 1) return p[0];
        if (bytes == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; }
        { uint32_t v; std::memcpy(&v, p, 4); return v; }
    }
}
} // namespace

bool load_ply(const std::string &path, HostMesh &mesh, std::string &error) {
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) {
        error = "Cannot open " + path;
        return false;
    }
    mesh = HostMesh();
    std::string line;
    if (!std::getline(in, line) || line.compare(0, 3, "ply") != 0) {
        error = "not a PLY file: " + path;
        return false;
    }
    bool ascii = false, known_format = false;
    struct Element {
        std::string name;
        long count;
        std::vector<PlyProp> props;
    };
    std::vector<Element> elements;
    while (std::getline(in, line)) {
        erase_crlf(line);
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "end_header") break;
        if (key == "format") {
            std::string f;
            ls >> f;
            ascii = f == "ascii";
            known_format = ascii || f == "binary_little_endian";
        } else if (key == "element") {
            Element e;
            ls >> e.name >> e.count;
            elements.push_back(e);
        } else if (key == "property" && !elements.empty()) {
            PlyProp p;
            std::string t;
            ls >> t;
            p.count_bytes = 0;
            bool ok = true;
            if (t == "list") {
                std::string ct, it;
                ls >> ct >> it;
                char ck;
                ok = ply_type(ct, p.count_bytes, ck) && ply_type(it, p.bytes, p.kind);
            } else {
                ok = ply_type(t, p.bytes, p.kind);
            }
            ls >> p.name;
            if (!ok) {
                error = "unknown PLY property type in " + path + ": " + line;
                return false;
            }
            elements.back().props.push_back(p);
        }
    }
    if (!known_format) {
        error = "PLY format must be ascii or binary_little_endian: " + path;
        return false;
    }
    std::vector<float> file_normals;
    auto read_value = [&](const PlyProp &p, bool count_field, double &out) -> bool {
        const int bytes = count_field ? p.count_bytes : p.bytes;
        const char kind = count_field ? 'u' : p.kind;
        if (ascii) return (bool)(in >> out);
        unsigned char buf[8];
        if (!in.read((char *)buf, bytes)) return false;
        out = ply_scalar(buf, bytes, kind);
        return true;
    };
    for (const Element &e : elements) {
        for (long i = 0; i < e.count; ++i) {
            float xyz[3] = {0, 0, 0}, nrm[3] = {0, 0, 0};
            bool have_n = false;
            for (const PlyProp &p : e.props) {
                if (p.count_bytes == 0) {
                    double v;
                    if (!read_value(p, false, v)) {
                        error = "PLY file ends early: " + path;
                        return false;
                    }
                    if (e.name == "vertex") {
                        if (p.name == "x") xyz[0] = (float)v;
                        if (p.name == "y") xyz[1] = (float)v;
                        if (p.name == "z") xyz[2] = (float)v;
                        if (p.name == "nx") nrm[0] = (float)v, have_n = true;
                        if (p.name == "ny") nrm[1] = (float)v;
                        if (p.name == "nz") nrm[2] = (float)v;
                    }
                } else {
                    double cnt;
                    if (!read_value(p, true, cnt) || cnt < 0 || cnt > 255) {
                        error = "bad PLY list in " + path;
                        return false;
                    }
                    std::vector<int32_t> ids;
                    for (int k = 0; k < (int)cnt; ++k) {
                        double v;
                        if (!read_value(p, false, v)) {
                            error = "PLY file ends early: " + path;
                            return false;
                        }
                        ids.push_back((int32_t)v);
                    }
                    if (e.name == "face" && (p.name == "vertex_indices" || p.name == "vertex_index")) {
                        if (ids.size() != 3) {
                            error = "only triangle faces are supported in PLY files: " + path;
                            return false;
                        }
                        mesh.face_vertex.insert(mesh.face_vertex.end(), ids.begin(), ids.end());
                    }
                }
            }
            if (e.name == "vertex") {
                mesh.vertices.insert(mesh.vertices.end(), xyz, xyz + 3);
                if (have_n) file_normals.insert(file_normals.end(), nrm, nrm + 3);
            }
        }
    }
    const int32_t nv = mesh.num_vertices();
    for (int32_t id : mesh.face_vertex)
        if (id < 0 || id >= nv) {
            error = "face references vertex " + std::to_string(id) + " of " + std::to_string(nv) + " in " + path;
            return false;
        }
    // materials: <name>.mtl beside the file, first material for every face
    std::string mtl = path;
    const size_t dot = mtl.find_last_of('.');
    if (dot != std::string::npos) mtl = mtl.substr(0, dot);
    mtl += ".mtl";
    int32_t material = -1;
    if (std::ifstream(mtl.c_str()).good() && load_mtl(mtl, mesh.materials)) material = 0;
    mesh.face_material.assign((size_t)mesh.num_faces(), material);
    const bool paired = (int32_t)(file_normals.size() / 3) == nv && nv > 0;
    if (paired) mesh.normals = file_normals;
    finish_mesh(mesh, paired);
    return load_material_textures(mesh, error);
}

bool write_ppm_binary(const std::string &path, const float *rgb, int width, int height) {
    std::FILE *out = std::fopen(path.c_str(), "wb");
    if (!out) return false;
    std::fprintf(out, "P6\n%d %d\n255\n", width, height);
    std::vector<unsigned char> row((size_t)width * 3);
    for (int j = 0; j < height; ++j) {
        for (int i = 0; i < width * 3; ++i) {
            const int v = std::min(255, (int)(255 * rgb[(size_t)j * width * 3 + i])); // ppmIO.hpp:203
            row[(size_t)i] = (unsigned char)std::max(v, 0);
        }
        std::fwrite(row.data(), 1, row.size(), out);
    }
    std::fclose(out);
    return true;
}

// The same two files from pixels that are quantised already (rtamd_render_ppm: bytes = min(255, (int)(255 * c)) with no channel
// negative): the text of utils/ppmIO.hpp:188-209 from a table of the 256 possible "value + blank" strings, or the bytes as they are.
bool write_ppm_bytes(const std::string &path, const unsigned char *pixels, int width, int height, bool binary) {
    std::FILE *out = std::fopen(path.c_str(), binary ? "wb" : "w");
    if (!out) return false;
    std::fprintf(out, "%s\n%d %d\n255\n", binary ? "P6" : "P3", width, height);
    if (binary) {
        std::fwrite(pixels, 1, (size_t)width * height * 3, out);
    } else {
        char text[256][5];
        int len[256];
        for (int v = 0; v < 256; ++v) {
            len[v] = std::snprintf(text[v], sizeof text[v], "%d", v);
            text[v][len[v]++] = ' ';
        }
        std::string buf;
        buf.reserve((size_t)width * 12 + 16);
        for (int j = 0; j < height; ++j) {
            buf.clear();
            const unsigned char *row = pixels + (size_t)j * width * 3;
            for (int i = 0; i < width * 3; ++i) buf.append(text[row[i]], (size_t)len[row[i]]);
            buf.push_back('\n');
            std::fwrite(buf.data(), 1, buf.size(), out);
        }
    }
    std::fclose(out);
    return true;
}

// utils/ppmIO.hpp:188-209: ASCII P3, each channel min(255, (int)(255 * c)), one text line per row.
bool write_ppm(const std::string &path, const float *rgb, int width, int height) {
    std::FILE *out = std::fopen(path.c_str(), "w");
    if (!out) return false;
    std::string buf;
    buf.reserve((size_t)width * 12 + 16);
    std::fprintf(out, "P3\n%d %d\n255\n", width, height);
    char tmp[48];
    for (int j = 0; j < height; ++j) {
        buf.clear();
        for (int i = 0; i < width; ++i) {
            const float *p = rgb + 3 * ((size_t)j * width + i);
            const int r = std::min(255, (int)(255 * p[0])), g = std::min(255, (int)(255 * p[1])), b = std::min(255, (int)(255 * p[2]));
            const int n = std::snprintf(tmp, sizeof tmp, "%d %d %d ", r, g, b);
            buf.append(tmp, (size_t)n);
        }
        buf.push_back('\n');
        std::fwrite(buf.data(), 1, buf.size(), out);
    }
    std::fclose(out);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Flycamera
// ------------------------------------------------------------------------------------------------
namespace {

// Eigen's fixed-size reductions associate a three-term sum as x + (y + z) (redux_novec_unroller); every product of the
// reference's camera code below goes through one, so the terms are grouped the same way here.
inline float sum3(float x, float y, float z) { return x + (y + z); }

// Matrix3f::inverse() as Eigen 3.3.7 computes it (Eigen/src/LU/InverseImpl.h, compute_inverse<..., 3>): cofactors, the
// determinant expanded along the FIRST COLUMN, one reciprocal, every cofactor times that reciprocal.
void mat3_inverse(const float *m /*row-major 3x4, linear part*/, float *inv /*3x3 row-major*/) {
    auto at = [&](int r, int c) { return m[4 * r + c]; };
    auto cofactor = [&](int i, int j) { // cofactor_3x3<i, j>
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return at(i1, j1) * at(i2, j2) - at(i1, j2) * at(i2, j1);
    };
    const float c0 = cofactor(0, 0), c1 = cofactor(1, 0), c2 = cofactor(2, 0);
    const float det = sum3(c0 * at(0, 0), c1 * at(1, 0), c2 * at(2, 0));
    const float invdet = 1.0f / det;
    inv[0] = c0 * invdet, inv[1] = c1 * invdet, inv[2] = c2 * invdet;
    for (int r = 1; r < 3; ++r)
        for (int c = 0; c < 3; ++c) inv[3 * r + c] = cofactor(c, r) * invdet;
}

// rotation by `angle` about unit axis `ax` (Rodrigues, as Eigen::AngleAxisf::toRotationMatrix)
void axis_angle(const float *ax, float angle, float *R) {
    const float s = std::sin(angle), c = std::cos(angle), t = 1.0f - c;
    const float x = ax[0], y = ax[1], z = ax[2];
    R[0] = t * x * x + c, R[1] = t * x * y - s * z, R[2] = t * x * z + s * y;
    R[3] = t * x * y + s * z, R[4] = t * y * y + c, R[5] = t * y * z - s * x;
    R[6] = t * x * z - s * y, R[7] = t * y * z + s * x, R[8] = t * z * z + c;
}
void mul3(const float *R, const float *v, float *o) {
    for (int r = 0; r < 3; ++r) o[r] = sum3(R[3 * r] * v[0], R[3 * r + 1] * v[1], R[3 * r + 2] * v[2]);
}
void norm3(float *v) {
    const float z = v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]);
    if (z > 0.0f) {
        const float s = std::sqrt(z);
        v[0] /= s, v[1] /= s, v[2] /= s;
    }
}

} // namespace

void Flycamera::reset() {
    translation_[0] = translation_[1] = translation_[2] = 0.0f;
    rot_x_ = rot_y_ = 0.0f;
    updateViewMatrix();
    reComputeViewMatrix();
}

void Flycamera::setPerspectiveMatrix(float fovy, float aspect, float, float) {
    fovy_ = fovy;
    aspect_ = aspect;
}

void Flycamera::setViewport(float w, float h) {
    viewport_[0] = viewport_[1] = 0.0f;
    viewport_[2] = w, viewport_[3] = h;
}

void Flycamera::translate(float dx, float dy, float dz) {
    const float up[3] = {0, 1, 0};
    float Y[9];
    axis_angle(up, rot_y_, Y);
    const float v[3] = {-dx, -dy, dz};
    float o[3];
    mul3(Y, v, o);
    for (int k = 0; k < 3; ++k) translation_[k] += o[k] * speed_;
}

void Flycamera::rotate(float angle_x, float angle_y) {
    rot_x_ += angle_x;
    rot_y_ += angle_y;
}

// view = Rotate(default_rotation = I) * Rotate(rotation_matrix) * Translate(0,0,-2) * Translate(translation)
void Flycamera::updateViewMatrix() {
    const float ux[3] = {1, 0, 0}, uy[3] = {0, 1, 0}, uz[3] = {0, 0, 1};
    float Y[9], X[9], rotX[3], rotY[3], rotZ[3], tmp[3];
    axis_angle(uy, rot_y_, Y);
    mul3(Y, ux, rotX);
    norm3(rotX);
    mul3(Y, uz, tmp);
    axis_angle(rotX, rot_x_, X);
    mul3(X, tmp, rotZ);
    norm3(rotZ);
    mul3(X, uy, rotY);
    norm3(rotY);
    const float R[9] = {rotX[0], rotX[1], rotX[2], rotY[0], rotY[1], rotY[2], rotZ[0], rotZ[1], rotZ[2]};
    // Affine3f::translate(v) adds linear * v to the translation: once for the default translation (0, 0, -2), once for
    // the user's (utils/flycamera.hpp:189-190) -- two roundings, not one product with the summed vector
    const float d[3] = {0.0f, 0.0f, -2.0f};
    float Ld[3], Lt[3];
    mul3(R, d, Ld);
    mul3(R, translation_, Lt);
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) view_[4 * r + k] = R[3 * r + k];
        view_[4 * r + 3] = (0.0f + Ld[r]) + Lt[r];
    }
}

void Flycamera::reComputeViewMatrix() {
    float inv[9];
    mat3_inverse(view_, inv);
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) inv_view_[4 * r + k] = inv[3 * r + k];
        // -(L^-1 * t), Transform::inverse (Eigen/src/Geometry/Transform.h)
        inv_view_[4 * r + 3] = -sum3(inv[3 * r] * view_[3], inv[3 * r + 1] * view_[7], inv[3 * r + 2] * view_[11]);
    }
}

std::array<float, 3> Flycamera::getCenter() const {
    float inv[9];
    mat3_inverse(view_, inv);
    std::array<float, 3> c;
    for (int r = 0; r < 3; ++r) c[r] = sum3(inv[3 * r] * -view_[3], inv[3 * r + 1] * -view_[7], inv[3 * r + 2] * -view_[11]); // camera.hpp:120
    return c;
}

void Flycamera::fill(rtamd_camera &c) const {
    const std::array<float, 3> o = getCenter();
    // +0.0 so that a negative zero from the sign flips above never reaches the ray origin
    for (int k = 0; k < 3; ++k) c.origin[k] = o[k] + 0.0f;
    for (int k = 0; k < 12; ++k) c.inv_view[k] = inv_view_[k] + 0.0f;
    std::memcpy(c.viewport, viewport_, sizeof c.viewport);
    c.aspect_ratio = aspect_;
    c.fovy = fovy_;
}

// ------------------------------------------------------------------------------------------------
// Flyscene
// ------------------------------------------------------------------------------------------------
Flyscene::~Flyscene() {
    if (pixel_bytes_) rtamd_host_free(pixel_bytes_);
    if (scene_) rtamd_scene_destroy(scene_);
}

void Flyscene::loadLightsAndSpheres() {
    if (lights_defaulted) return;
    lights_defaulted = true;
    lights.insert(lights.begin(), Light{{-1.0f, 1.0f, 1.0f}, {1.0f, 1.0f, 1.0f}}); // src/flyscene.cpp:9
}

void Flyscene::precomputeData() {
    const int32_t nv = mesh.num_vertices(), nf = mesh.num_faces();
    // Mesh::loadVertices (mesh.hpp:610-628) + Model::normalizeModelMatrix (model.hpp:169-173)
    float s = 1.0f, c[3] = {0, 0, 0};
    if (nv > 0) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) lo[a] = hi[a] = mesh.vertices[a];
        for (int32_t i = 0; i < nv; ++i)
            for (int a = 0; a < 3; ++a) {
                hi[a] = std::max(mesh.vertices[3 * (size_t)i + a], hi[a]);
                lo[a] = std::min(mesh.vertices[3 * (size_t)i + a], lo[a]);
            }
        const float ext = std::max(std::max(std::fabs(hi[0] - lo[0]), std::fabs(hi[1] - lo[1])), std::fabs(hi[2] - lo[2]));
        s = (float)(1.0 / (double)ext);
        for (int a = 0; a < 3; ++a) c[a] = (float)((double)(hi[a] + lo[a]) / 2.0);
    }
    std::memset(model_matrix, 0, sizeof model_matrix);
    if (normalize_model) {
        model_matrix[0] = model_matrix[5] = model_matrix[10] = s;
        model_matrix[3] = s * -c[0], model_matrix[7] = s * -c[1], model_matrix[11] = s * -c[2];
    } else {
        model_matrix[0] = model_matrix[5] = model_matrix[10] = 1.0f;
    }
    // src/flyscene.cpp:165-219.  Entries of vertices never referenced by a face stay zero.
    P_vertices.assign(3 * (size_t)nv, 0.0f);
    P_normals.assign(3 * (size_t)nv, 0.0f);
    P_face_distance.assign((size_t)nf, 0.0f);
    for (int32_t f = 0; f < nf; ++f) {
        const int32_t *id = &mesh.face_vertex[3 * (size_t)f];
        for (int k = 0; k < 3; ++k) {
            const float *v = &mesh.vertices[3 * (size_t)id[k]];
            float *o = &P_vertices[3 * (size_t)id[k]];
            // (shapeModelMatrix * Vector4f(v, 1)).head(3): one multiply and one add per coordinate
            o[0] = model_matrix[0] * v[0] + model_matrix[3];
            o[1] = model_matrix[5] * v[1] + model_matrix[7];
            o[2] = model_matrix[10] * v[2] + model_matrix[11];
            put(&P_normals[3 * (size_t)id[k]], unit(f3(&mesh.normals[3 * (size_t)id[k]])));
        }
        P_face_distance[f] = dot(f3(&mesh.face_normal[3 * (size_t)f]), f3(&P_vertices[3 * (size_t)id[0]])); // :209
    }
    plain_materials.clear();
    for (const NamedMaterial &nm : mesh.materials) plain_materials.push_back(nm.m);
}

void Flyscene::precomputeLights() {
    const int spl = (hard_shadow ? 1 : 0) + soft_points;
    light_samples.assign(3 * lights.size() * (size_t)spl, 0.0f);
    sampled_lights = lights.size();
    for (size_t i = 0; i < lights.size(); ++i) {
        float *out = &light_samples[3 * i * (size_t)spl];
        const float *lp = lights[i].position;
        int k = 0;
        if (hard_shadow) { // :228-233
            out[0] = lp[0], out[1] = lp[1], out[2] = lp[2];
            k = 1;
        }
        for (int n = 0; n < soft_points; ++n, ++k) { // :240-247
            const float theta = (2.0f * (float)n * (float)M_PI) / soft_points;
            const float phi = std::acos(1 - 2 * ((float)n / (soft_points * (float)M_PI)));
            const float x = soft_radius * std::sin(phi) * std::cos(theta);
            const float y = soft_radius * std::sin(phi) * std::sin(theta);
            const float z = soft_radius * std::cos(phi);
            out[3 * k] = lp[0] + x, out[3 * k + 1] = lp[1] + y, out[3 * k + 2] = lp[2] + z;
        }
    }
}

int Flyscene::upload() {
    if (scene_) {
        rtamd_scene_destroy(scene_);
        scene_ = nullptr;
    }
    std::vector<float> lpos, lcol;
    for (const Light &l : lights) {
        lpos.insert(lpos.end(), l.position, l.position + 3);
        lcol.insert(lcol.end(), l.color, l.color + 3);
    }
    rtamd_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.vertices = P_vertices.data();
    d.normals = P_normals.data();
    d.num_vertices = mesh.num_vertices();
    d.face_vertex_ids = mesh.face_vertex.data();
    d.face_normals = mesh.face_normal.data();
    d.face_origin_distance = P_face_distance.data();
    d.face_material_ids = mesh.face_material.data();
    d.num_faces = mesh.num_faces();
    d.materials = plain_materials.data();
    d.num_materials = (int32_t)plain_materials.size();
    d.light_positions = lpos.data();
    d.light_colors = lcol.data();
    d.num_lights = (int32_t)lights.size();
    d.num_unsampled_lights = (int32_t)(lights.size() - std::min(sampled_lights, lights.size())); // lights of addLight()
    d.light_samples = light_samples.data();
    d.samples_per_light = (hard_shadow ? 1 : 0) + soft_points;
    d.spheres = spheres.data();
    d.num_spheres = (int32_t)spheres.size();
    std::vector<rtamd_texture> tex;
    for (const HostTexture &t : mesh.textures) tex.push_back(rtamd_texture{t.width, t.height, t.rgb.data()});
    if (!mesh.face_texcoord.empty() && !tex.empty() && mesh.material_texture.size() == plain_materials.size()) {
        d.face_texcoords = mesh.face_texcoord.data();
        d.material_textures = mesh.material_texture.data();
        d.textures = tex.data();
        d.num_textures = (int32_t)tex.size();
    }
    return rtamd_scene_create(&d, &build, &scene_);
}

int Flyscene::initialize(int width, int height, bool do_upload) {
    loadLightsAndSpheres();
    flycamera.setPerspectiveMatrix(60.0f, (float)width / (float)height, 0.1f, 100.0f); // :35
    flycamera.setViewport((float)width, (float)height);                                  // :36
    if (!mesh_from_arrays) {
        std::string err;
        const bool is_ply = model_path.size() >= 4 && (model_path.compare(model_path.size() - 4, 4, ".ply") == 0 ||
                                                        model_path.compare(model_path.size() - 4, 4, ".PLY") == 0);
        if (!(is_ply ? load_ply(model_path, mesh, err) : load_obj(model_path, mesh, err))) {
            std::fprintf(stderr, "%s\n", err.c_str());
            return RTAMD_ERR_INVALID;
        }
    }
    if (params.width == 0) rtamd_render_params_default(&params, width, height);
    params.width = width, params.height = height;
    precomputeData();
    precomputeLights();
    initialized_ = true;
    return do_upload ? upload() : RTAMD_OK;
}

// src/flyscene.hpp:76: `lights.push_back(...)` and nothing else.  precomputeLights is NOT run again (the reference
// calls it once, from initialize, src/flyscene.cpp:117), so the new light has no shadow sample points: it is shaded
// (:505) but getLightIntensity (:880) never asks whether it is occluded.  The uploaded scene keeps its hierarchy and
// shadow-ray structures; only its light list is replaced.
int Flyscene::addLight() {
    const std::array<float, 3> c = flycamera.getCenter();
    lights.push_back(Light{{c[0], c[1], c[2]}, {1.0f, 1.0f, 1.0f}});
    if (!initialized_ || !scene_) return RTAMD_OK;
    std::vector<float> lpos, lcol;
    for (const Light &l : lights) {
        lpos.insert(lpos.end(), l.position, l.position + 3);
        lcol.insert(lcol.end(), l.color, l.color + 3);
    }
    return rtamd_scene_set_lights(scene_, lpos.data(), lcol.data(), (int32_t)lights.size());
}

int Flyscene::render(float *rgb, int32_t *face_id, rtamd_stats *stats) {
    if (!scene_) return RTAMD_ERR_INVALID;
    flycamera.reComputeViewMatrix(); // tracePixels does this first, src/flyscene.cpp:806
    rtamd_camera cam;
    flycamera.fill(cam);
    return rtamd_render(scene_, &cam, &params, rgb, face_id, stats);
}

// src/flyscene.cpp:381-459.  All the reference keeps of a frame is result.ppm, so the frame comes back from the device as
// the file's 8-bit pixels (rtamd_render_ppm: a quarter of the bytes of pixel_data over the host link) into a page-locked
// buffer this object keeps; only a frame with a channel the bytes cannot express (negative, not a number) is fetched as
// floats, as the reference holds it, and written from those.
int Flyscene::raytraceScene() {
    const std::array<int, 2> size = flycamera.getViewportSize(); // :394
    if (size[0] != params.width || size[1] != params.height) return RTAMD_ERR_INVALID;
    if (!scene_) return RTAMD_ERR_INVALID;
    rtamd_render_params full = params;
    full.num_parts = 1, full.part = 0;
    const size_t nbytes = (size_t)size[0] * (size_t)size[1] * 3;
    if (pixel_bytes_size_ < nbytes) {
        if (pixel_bytes_) rtamd_host_free(pixel_bytes_);
        pixel_bytes_ = nullptr, pixel_bytes_size_ = 0;
        const int rc = rtamd_host_alloc(&pixel_bytes_, (uint64_t)nbytes);
        if (rc != RTAMD_OK) return rc;
        pixel_bytes_size_ = nbytes;
    }
    flycamera.reComputeViewMatrix(); // tracePixels does this first, src/flyscene.cpp:806
    rtamd_camera cam;
    flycamera.fill(cam);
    int32_t exact = 1;
    int rc = rtamd_render_ppm(scene_, &cam, &full, (uint8_t *)pixel_bytes_, &exact, nullptr);
    if (rc != RTAMD_OK) return rc;
    auto ends_with = [&](const char *suffix) {
        const size_t n = std::strlen(suffix);
        return output_path.size() >= n && output_path.compare(output_path.size() - n, n, suffix) == 0;
    };
    const bool binary = ends_with(".p6.ppm") || ends_with(".pnm");
    if (exact) {
        if (!output_path.empty() && !write_ppm_bytes(output_path, (const unsigned char *)pixel_bytes_, size[0], size[1], binary)) return RTAMD_ERR_INVALID;
        return RTAMD_OK;
    }
    // pixel_data (src/flyscene.cpp:399-403) as floats, for the writer that prints what the reference prints of them
    std::vector<float> pixel_data((size_t)size[0] * (size_t)size[1] * 3);
    rc = rtamd_render(scene_, &cam, &full, pixel_data.data(), nullptr, nullptr);
    if (rc != RTAMD_OK) return rc;
    if (!output_path.empty() &&
        !(binary ? write_ppm_binary(output_path, pixel_data.data(), size[0], size[1]) : write_ppm(output_path, pixel_data.data(), size[0], size[1])))
        return RTAMD_ERR_INVALID;
    return RTAMD_OK;
}

} // namespace rtamd

// ------------------------------------------------------------------------------------------------
// Flat C view
// ------------------------------------------------------------------------------------------------
using rtamd::Flyscene;

struct rtfly {
    Flyscene fly;
};

extern "C" {

int rtfly_create(rtfly **out) {
    if (!out) return RTAMD_ERR_INVALID;
    *out = new rtfly();
    return RTAMD_OK;
}
int rtfly_destroy(rtfly *f) {
    delete f;
    return RTAMD_OK;
}
int rtfly_set_model(rtfly *f, const char *p) {
    if (!f || !p) return RTAMD_ERR_INVALID;
    f->fly.model_path = p;
    f->fly.mesh_from_arrays = false;
    return RTAMD_OK;
}
int rtfly_set_model_arrays(rtfly *f, const float *v, const float *n, int32_t nv, const int32_t *faces, const int32_t *fm, int32_t nf,
                           const rtamd_material *mats, int32_t nm) {
    if (!f || !v || !faces || nv < 0 || nf < 0 || nm < 0) return RTAMD_ERR_INVALID;
    for (int64_t i = 0; i < 3 * (int64_t)nf; ++i)
        if (faces[i] < 0 || faces[i] >= nv) return RTAMD_ERR_INVALID;
    rtamd::HostMesh &m = f->fly.mesh;
    m = rtamd::HostMesh();
    m.vertices.assign(v, v + 3 * (size_t)nv);
    if (n) m.normals.assign(n, n + 3 * (size_t)nv);
    m.face_vertex.assign(faces, faces + 3 * (size_t)nf);
    m.face_material.assign((size_t)nf, -1);
    if (fm) m.face_material.assign(fm, fm + nf);
    for (int32_t i = 0; i < nm; ++i) m.materials.push_back(rtamd::NamedMaterial{std::string(), mats[i], std::string()});
    rtamd::finish_mesh(m, n != nullptr);
    f->fly.mesh_from_arrays = true;
    return RTAMD_OK;
}
int rtfly_set_texture_arrays(rtfly *f, const float *face_texcoords, const int32_t *material_textures, const rtamd_texture *textures,
                             int32_t num_textures) {
    if (!f || num_textures < 0 || (num_textures > 0 && (!textures || !material_textures || !face_texcoords))) return RTAMD_ERR_INVALID;
    rtamd::HostMesh &m = f->fly.mesh;
    m.face_texcoord.clear();
    m.textures.clear();
    m.material_texture.assign(m.materials.size(), -1);
    if (num_textures == 0) return RTAMD_OK;
    m.face_texcoord.assign(face_texcoords, face_texcoords + 6 * (size_t)m.num_faces());
    for (int32_t i = 0; i < num_textures; ++i) {
        if (textures[i].width <= 0 || textures[i].height <= 0 || !textures[i].rgb) return RTAMD_ERR_INVALID;
        rtamd::HostTexture t;
        t.width = textures[i].width, t.height = textures[i].height;
        t.rgb.assign(textures[i].rgb, textures[i].rgb + (size_t)t.width * t.height * 3);
        m.textures.push_back(std::move(t));
    }
    for (size_t i = 0; i < m.materials.size(); ++i) {
        const int32_t t = material_textures[i];
        if (t < -1 || t >= num_textures) return RTAMD_ERR_INVALID;
        m.material_texture[i] = t;
    }
    return RTAMD_OK;
}
int rtfly_clear_lights(rtfly *f) {
    if (!f) return RTAMD_ERR_INVALID;
    f->fly.lights.clear();
    f->fly.lights_defaulted = true;
    return RTAMD_OK;
}
int rtfly_add_light_at(rtfly *f, const float *p, const float *c) {
    if (!f || !p) return RTAMD_ERR_INVALID;
    rtamd::Light l{{p[0], p[1], p[2]}, {1.0f, 1.0f, 1.0f}};
    if (c) l.color[0] = c[0], l.color[1] = c[1], l.color[2] = c[2];
    f->fly.lights.push_back(l);
    return RTAMD_OK;
}
int rtfly_set_shadow_sampling(rtfly *f, int32_t hard, int32_t soft, float radius) {
    if (!f || soft < 0 || (!hard && soft == 0)) return RTAMD_ERR_INVALID;
    f->fly.hard_shadow = hard != 0, f->fly.soft_points = soft, f->fly.soft_radius = radius;
    return RTAMD_OK;
}
int rtfly_add_sphere(rtfly *f, const float *c, float r) {
    if (!f || !c) return RTAMD_ERR_INVALID;
    f->fly.spheres.push_back(rtamd_sphere{{c[0], c[1], c[2]}, r});
    return RTAMD_OK;
}
int rtfly_set_normalize_model(rtfly *f, int32_t on) {
    if (!f) return RTAMD_ERR_INVALID;
    f->fly.normalize_model = on != 0;
    return RTAMD_OK;
}
int rtfly_set_output(rtfly *f, const char *p) {
    if (!f || !p) return RTAMD_ERR_INVALID;
    f->fly.output_path = p;
    return RTAMD_OK;
}
int rtfly_set_build_params(rtfly *f, const rtamd_build_params *b) {
    if (!f || !b) return RTAMD_ERR_INVALID;
    f->fly.build = *b;
    return RTAMD_OK;
}
int rtfly_initialize(rtfly *f, int32_t w, int32_t h, int32_t upload) {
    if (!f || w <= 0 || h <= 0) return RTAMD_ERR_INVALID;
    return f->fly.initialize(w, h, upload != 0);
}
int rtfly_add_light(rtfly *f) { return f ? f->fly.addLight() : RTAMD_ERR_INVALID; }
int rtfly_camera_reset(rtfly *f) {
    if (!f) return RTAMD_ERR_INVALID;
    f->fly.getCamera()->reset();
    return RTAMD_OK;
}
int rtfly_camera_translate(rtfly *f, float dx, float dy, float dz) {
    if (!f) return RTAMD_ERR_INVALID;
    f->fly.getCamera()->translate(dx, dy, dz);
    return RTAMD_OK;
}
int rtfly_camera_rotate(rtfly *f, float ax, float ay) {
    if (!f) return RTAMD_ERR_INVALID;
    f->fly.getCamera()->rotate(ax, ay);
    return RTAMD_OK;
}
int rtfly_camera_update_view(rtfly *f) {
    if (!f) return RTAMD_ERR_INVALID;
    f->fly.getCamera()->updateViewMatrix();
    return RTAMD_OK;
}
int rtfly_get_camera(rtfly *f, rtamd_camera *cam) {
    if (!f || !cam) return RTAMD_ERR_INVALID;
    f->fly.getCamera()->reComputeViewMatrix();
    f->fly.getCamera()->fill(*cam);
    return RTAMD_OK;
}
rtamd_render_params *rtfly_params(rtfly *f) { return f ? &f->fly.params : nullptr; }
int rtfly_raytrace_scene(rtfly *f) { return f ? f->fly.raytraceScene() : RTAMD_ERR_INVALID; }
int rtfly_raytrace_to(rtfly *f, float *rgb, int32_t *face_id, rtamd_stats *stats) {
    if (!f || !rgb) return RTAMD_ERR_INVALID;
    return f->fly.render(rgb, face_id, stats);
}
rtamd_scene *rtfly_scene(rtfly *f) { return f ? f->fly.scene() : nullptr; }

int rtfly_get_array(rtfly *f, const char *name, const void **data, int64_t *count) {
    if (!f || !name || !data || !count) return RTAMD_ERR_INVALID;
    Flyscene &s = f->fly;
    const std::string n = name;
    auto set = [&](const void *p, size_t c) {
        *data = p;
        *count = (int64_t)c;
        return RTAMD_OK;
    };
    if (n == "raw_vertices") return set(s.mesh.vertices.data(), s.mesh.vertices.size());
    if (n == "raw_normals") return set(s.mesh.normals.data(), s.mesh.normals.size());
    if (n == "vertices") return set(s.P_vertices.data(), s.P_vertices.size());
    if (n == "normals") return set(s.P_normals.data(), s.P_normals.size());
    if (n == "face_normals") return set(s.mesh.face_normal.data(), s.mesh.face_normal.size());
    if (n == "face_origin_distance") return set(s.P_face_distance.data(), s.P_face_distance.size());
    if (n == "light_samples") return set(s.light_samples.data(), s.light_samples.size());
    if (n == "materials") return set(s.plain_materials.data(), s.plain_materials.size() * 10);
    if (n == "model_matrix") return set(s.model_matrix, 12);
    if (n == "face_vertex_ids") return set(s.mesh.face_vertex.data(), s.mesh.face_vertex.size());
    if (n == "face_material_ids") return set(s.mesh.face_material.data(), s.mesh.face_material.size());
    if (n == "face_texcoords") return set(s.mesh.face_texcoord.data(), s.mesh.face_texcoord.size());
    if (n == "material_textures") return set(s.mesh.material_texture.data(), s.mesh.material_texture.size());
    if (n.compare(0, 8, "texture:") == 0) { // "texture:<k>" -> texels; "texture_size:<k>" below -> (width, height)
        const size_t k = (size_t)std::strtoul(n.c_str() + 8, nullptr, 10);
        if (k >= s.mesh.textures.size()) return RTAMD_ERR_INVALID;
        return set(s.mesh.textures[k].rgb.data(), s.mesh.textures[k].rgb.size());
    }
    if (n.compare(0, 13, "texture_size:") == 0) {
        const size_t k = (size_t)std::strtoul(n.c_str() + 13, nullptr, 10);
        if (k >= s.mesh.textures.size()) return RTAMD_ERR_INVALID;
        return set(&s.mesh.textures[k].width, 2);
    }
    return RTAMD_ERR_INVALID;
}

int rtfly_write_ppm_binary(const char *path, const float *rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) return RTAMD_ERR_INVALID;
    return rtamd::write_ppm_binary(path, rgb, w, h) ? RTAMD_OK : RTAMD_ERR_INVALID;
}

int rtfly_write_ppm(const char *path, const float *rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) return RTAMD_ERR_INVALID;
    return rtamd::write_ppm(path, rgb, w, h) ? RTAMD_OK : RTAMD_ERR_INVALID;
}

} // extern "C"
