/*
 * CPU oracle for the Flyscene ray-tracing pass.  TEST INFRASTRUCTURE ONLY -- see oracle.h.
 *
 * Every function names the reference lines it restates (paths relative to the reference root).
 * Known deviations, none of which changes a pixel: no GL objects inside BVH nodes
 * (boundingBox.h:36), no hitByRay stores (boundingBox.h:297,305), arrays are zero-initialised
 * where the reference leaves them uninitialised (flyscene.cpp:169-176).
 * Merge residue in the reference is resolved as follows: the undeclared `faceArea` division
 * (flyscene.cpp:499) is dropped (the next line re-normalises), the light direction is
 * `lightPosition.first - hitPosition` (:507), and the accumulated colour is the clamped sum
 * without the light colour (:554-564), which is what both variants give for white lights.
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "oracle_vec.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------ */

typedef struct orc_node {
    v3 vmin, vmax;
    int *faces;
    int nfaces;
    struct orc_node *child[2]; /* {lower, upper}; NULL,NULL for a leaf */
} orc_node;

typedef struct {
    char name[256];
    orc_material m;
    char tex_file[1024]; /* Mtl::diffuse_tex_filename, "" = none */
} named_material;

typedef struct {
    int w, h;
    float *rgb;          /* texturedatas[filename]: w*h*3, rows flipped as loadPPMImageData leaves them */
    char file[1024];
} orc_tex;

struct orc_scene {
    /* S1: mesh as Tucano::Mesh holds it */
    int nv, nf;
    float *raw_v;  /* nv*3 file coordinates (w = 1 implied) */
    float *raw_n;  /* nv*3 per-vertex normals as stored by the importer */
    int *fvid;     /* nf*3 */
    float *fnormal; /* nf*3 Face::normal, mesh.hpp:475-477 */
    int *fmat;     /* nf */
    float *ftex;   /* nf*6 Face::texcoord (u,v per corner), NULL when the mesh has no texture coordinates */
    int ntex;
    orc_tex *tex;
    int *mat_tex;  /* nmat: index into tex, -1 = material without map_kd */
    int nmat;
    named_material *mats;
    orc_material *mats_plain;
    /* S3 */
    float norm_scale;
    v3 center;
    float model[12];
    /* S4: PrecomputedData */
    float *P_v, *P_n, *P_D;
    /* S5 */
    int nl, spl;   /* lights that have a set of shadow sample points (PrecomputedData::lights), points per set */
    int nl_shade;  /* entries of Flyscene::lights: nl + the lights addLight pushed after precomputeLights ran */
    float *light_pos, *light_col, *samples;
    /* spheres */
    int ns;
    float *spheres;
    /* B1/B2 */
    orc_node *root;
    int node_count, leaf_count, depth;
};

/* ---------------------------------- small dynamic arrays ---------------------------------- */

typedef struct {
    int *p;
    size_t n, cap;
} ivec;
static void ivec_push(ivec *v, int x) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 64;
        v->p = (int *)realloc(v->p, v->cap * sizeof(int));
    }
    v->p[v->n++] = x;
}
typedef struct {
    float *p;
    size_t n, cap;
} fvec;
static void fvec_push3(fvec *v, float a, float b, float c) {
    if (v->n + 3 > v->cap) {
        v->cap = v->cap ? v->cap * 2 : 192;
        v->p = (float *)realloc(v->p, v->cap * sizeof(float));
    }
    v->p[v->n++] = a;
    v->p[v->n++] = b;
    v->p[v->n++] = c;
}

/* ------------------------------------ S2: MTL reading ------------------------------------- */

static void strip_crlf(char *s) {
    /* std::remove of '\n' and '\r' anywhere in the string (mtlIO.hpp:86-87, objimporter.hpp:164-165) */
    char *w = s;
    for (char *r = s; *r; ++r)
        if (*r != '\n' && *r != '\r') *w++ = *r;
    *w = 0;
}

static void material_defaults(orc_material *m) {
    /* mtl.hpp:22-31; Vector3f built from double literals */
    for (int i = 0; i < 3; ++i) {
        m->ka[i] = (float)0.3;
        m->kd[i] = (float)0.5;
        m->ks[i] = (float)1.0;
    }
    m->shininess = 10;
}

/* mtlIO.hpp:49-140.  Tokens are split on single spaces only (std::getline(ss, s, ' ')), so
 * doubled spaces give empty tokens and tabs are not separators. */
static int load_mtl(const char *filename, named_material **mats, int *nmat) {
    FILE *in = fopen(filename, "r");
    if (!in) {
        fprintf(stderr, "Cannot open %s\n", filename);
        return 0;
    }
    char *line = NULL;
    size_t cap = 0;
    ssize_t len;
    while ((len = getline(&line, &cap, in)) >= 0) {
        if (len > 0 && line[len - 1] == '\n') line[--len] = 0;
        if (len == 0) continue;
        /* split on ' ' */
        char *tok[64];
        int nt = 0;
        char *p = line;
        while (nt < 64) {
            tok[nt++] = p;
            char *sp = strchr(p, ' ');
            if (!sp) break;
            *sp = 0;
            p = sp + 1;
            if (*p == 0) break; /* getline at end of stream yields no further token */
        }
        const char *t1 = nt > 1 ? tok[1] : "";
        const char *t2 = nt > 2 ? tok[2] : "";
        const char *t3 = nt > 3 ? tok[3] : "";
        if (strcmp(tok[0], "#") == 0) continue;
        if (strcmp(tok[0], "newmtl") == 0) {
            *mats = (named_material *)realloc(*mats, (size_t)(*nmat + 1) * sizeof(named_material));
            named_material *m = &(*mats)[(*nmat)++];
            memset(m, 0, sizeof(*m));
            material_defaults(&m->m);
            char nm[256];
            snprintf(nm, sizeof nm, "%s", t1);
            strip_crlf(nm);
            snprintf(m->name, sizeof m->name, "%s", nm);
        } else if (*nmat > 0) {
            orc_material *m = &(*mats)[*nmat - 1].m;
            if (strcmp(tok[0], "Ns") == 0) {
                m->shininess = (float)atof(t1);
            } else if (strcmp(tok[0], "Ka") == 0) {
                m->ka[0] = (float)atof(t1), m->ka[1] = (float)atof(t2), m->ka[2] = (float)atof(t3);
            } else if (strcmp(tok[0], "Kd") == 0) {
                m->kd[0] = (float)atof(t1), m->kd[1] = (float)atof(t2), m->kd[2] = (float)atof(t3);
            } else if (strcmp(tok[0], "Ks") == 0) {
                m->ks[0] = (float)atof(t1), m->ks[1] = (float)atof(t2), m->ks[2] = (float)atof(t3);
            }
            else if (strcmp(tok[0], "map_kd") == 0) { /* lower case only, mtlIO.hpp:122-125 */
                char dir[1024];
                const char *a = strrchr(filename, '/');
                const char *b = strrchr(filename, '\\');
                const char *cut = a > b ? a : b;
                size_t dl = cut ? (size_t)(cut - filename) + 1 : 0;
                if (dl >= sizeof dir) dl = sizeof dir - 1;
                memcpy(dir, filename, dl);
                dir[dl] = 0;
                named_material *nm = &(*mats)[*nmat - 1];
                snprintf(nm->tex_file, sizeof nm->tex_file, "%s%s", dir, t1);
            }
            /* Ni, d, illum: not read on the traced path */
        }
    }
    free(line);
    fclose(in);
    if (*nmat == 0) { /* mtlIO.hpp:130-134 */
        *mats = (named_material *)realloc(*mats, sizeof(named_material));
        memset(*mats, 0, sizeof(named_material));
        material_defaults(&(*mats)[0].m);
        *nmat = 1;
    }
    return 1;
}

/* ---------------------------------- texture (PPM) reading ---------------------------------- */

/* ImageImporter::loadPPMImageData, ppmIO.hpp:36-86: header word, width, height, maximum, then
 * value/maximum for every number, rows flipped.  Returns 0 when the file cannot be read as that (the
 * reference then indexes empty vectors). */
static int load_ppm_texture(const char *filename, orc_tex *t) {
    char fn[1024];
    snprintf(fn, sizeof fn, "%s", filename);
    strip_crlf(fn); /* :40-41 */
    memset(t, 0, sizeof *t);
    snprintf(t->file, sizeof t->file, "%s", fn);
    FILE *in = fopen(fn, "r");
    if (!in) {
        fprintf(stderr, "Cannot open %s\n", fn);
        return 0;
    }
    char header[64];
    float maxv = 0;
    int ok = fscanf(in, "%63s", header) == 1 && fscanf(in, "%d %d", &t->w, &t->h) == 2 && fscanf(in, "%f", &maxv) == 1;
    if (!ok || t->w <= 0 || t->h <= 0) {
        fclose(in);
        t->w = t->h = 0;
        return 0;
    }
    const size_t n = (size_t)t->w * (size_t)t->h * 3;
    float *data = (float *)malloc(n * sizeof(float));
    size_t got = 0;
    float v;
    while (got < n && fscanf(in, "%f", &v) == 1) data[got++] = v / maxv; /* :57-60 */
    fclose(in);
    if (got < n) {
        free(data);
        t->w = t->h = 0;
        return 0;
    }
    t->rgb = (float *)malloc(n * sizeof(float));
    for (int j = t->h - 1, r = 0; j >= 0; --j, ++r) /* :68-77 */
        memcpy(t->rgb + (size_t)r * t->w * 3, data + (size_t)j * t->w * 3, (size_t)t->w * 3 * sizeof(float));
    free(data);
    return 1;
}

/* Flyscene::initialize, flyscene.cpp:52-71: one texture per distinct file name (std::map::insert keeps the
 * first); mat_tex[i] = its index. */
static void load_material_textures(orc_scene *s) {
    s->mat_tex = (int *)malloc((size_t)(s->nmat > 0 ? s->nmat : 1) * sizeof(int));
    for (int i = 0; i < s->nmat; ++i) {
        s->mat_tex[i] = -1;
        if (!s->mats[i].tex_file[0]) continue;
        char fn[1024];
        snprintf(fn, sizeof fn, "%s", s->mats[i].tex_file);
        strip_crlf(fn);
        for (int k = 0; k < s->ntex; ++k)
            if (strcmp(s->tex[k].file, fn) == 0) s->mat_tex[i] = k;
        if (s->mat_tex[i] >= 0) continue;
        s->tex = (orc_tex *)realloc(s->tex, (size_t)(s->ntex + 1) * sizeof(orc_tex));
        load_ppm_texture(fn, &s->tex[s->ntex]);
        s->mat_tex[i] = s->ntex++;
    }
}

/* ------------------------------------ S1: OBJ reading ------------------------------------- */

static void face_normal_raw(const float *v, int i0, int i1, int i2, float *out) {
    /* mesh.hpp:475-477 (same arithmetic as objimporter.hpp:93-95) */
    v3 p0 = v3_make(v[3 * i0], v[3 * i0 + 1], v[3 * i0 + 2]);
    v3 p1 = v3_make(v[3 * i1], v[3 * i1 + 1], v[3 * i1 + 2]);
    v3 p2 = v3_make(v[3 * i2], v[3 * i2 + 1], v[3 * i2 + 2]);
    v3 e1 = v3_normalized(v3_sub(p2, p0));
    v3 e0 = v3_normalized(v3_sub(p1, p0));
    v3 n = v3_normalized(v3_cross(e0, e1));
    out[0] = n.x, out[1] = n.y, out[2] = n.z;
}

/* objimporter.hpp:81-106 */
static void compute_vertex_normals(orc_scene *s) {
    memset(s->raw_n, 0, (size_t)s->nv * 3 * sizeof(float));
    for (int f = 0; f < s->nf; ++f) {
        float n[3];
        face_normal_raw(s->raw_v, s->fvid[3 * f], s->fvid[3 * f + 1], s->fvid[3 * f + 2], n);
        for (int k = 0; k < 3; ++k) {
            float *d = s->raw_n + 3 * s->fvid[3 * f + k];
            d[0] += n[0], d[1] += n[1], d[2] += n[2];
        }
    }
    for (int i = 0; i < s->nv; ++i) {
        v3 n = v3_normalized(v3_make(s->raw_n[3 * i], s->raw_n[3 * i + 1], s->raw_n[3 * i + 2]));
        s->raw_n[3 * i] = n.x, s->raw_n[3 * i + 1] = n.y, s->raw_n[3 * i + 2] = n.z;
    }
}

static void finish_faces(orc_scene *s) {
    s->fnormal = (float *)calloc((size_t)s->nf * 3 + 1, sizeof(float));
    for (int f = 0; f < s->nf; ++f)
        face_normal_raw(s->raw_v, s->fvid[3 * f], s->fvid[3 * f + 1], s->fvid[3 * f + 2], s->fnormal + 3 * f);
}

static void dirname_of(const char *path, char *out, size_t n) {
    /* getPathName, objimporter.hpp:44-48 */
    const char *a = strrchr(path, '/');
    const char *b = strrchr(path, '\\');
    const char *cut = a > b ? a : b;
    size_t len = cut ? (size_t)(cut - path) + 1 : 0;
    if (len >= n) len = n - 1;
    memcpy(out, path, len);
    out[len] = 0;
}

orc_scene *orc_scene_load_obj(const char *path) {
    FILE *in = fopen(path, "r");
    if (!in) {
        fprintf(stderr, "Cannot open %s\n", path);
        return NULL;
    }
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    char dir[1024];
    dirname_of(path, dir, sizeof dir);

    fvec vert = {0}, norm = {0};
    fvec vt = {0}; /* pushed three at a time: (u, v, 0) */
    /* groups: one index list per usemtl block, objimporter.hpp:144-148,168-177 */
    int ngroups = 1, gcap = 8;
    ivec *groups = (ivec *)calloc((size_t)gcap, sizeof(ivec));
    ivec *gtex = (ivec *)calloc((size_t)gcap, sizeof(ivec)); /* elementsTexIDs */
    int *gmat = (int *)malloc((size_t)gcap * sizeof(int));
    gmat[0] = -1;
    int current_mat = -1;

    char *line = NULL;
    size_t cap = 0;
    ssize_t len;
    while ((len = getline(&line, &cap, in)) >= 0) {
        if (len > 0 && line[len - 1] == '\n') line[--len] = 0;
        if (strncmp(line, "mtllib", 6) == 0 && len >= 7) {
            char fn[2048];
            snprintf(fn, sizeof fn, "%s%s", dir, line + 7);
            strip_crlf(fn);
            load_mtl(fn, &s->mats, &s->nmat);
        } else if (strncmp(line, "usemtl", 6) == 0 && len >= 7) {
            if (groups[ngroups - 1].n != 0) {
                if (ngroups == gcap) {
                    gcap *= 2;
                    groups = (ivec *)realloc(groups, (size_t)gcap * sizeof(ivec));
                    gtex = (ivec *)realloc(gtex, (size_t)gcap * sizeof(ivec));
                    gmat = (int *)realloc(gmat, (size_t)gcap * sizeof(int));
                }
                memset(&groups[ngroups], 0, sizeof(ivec));
                memset(&gtex[ngroups], 0, sizeof(ivec));
                gmat[ngroups] = -1;
                ++ngroups;
            }
            char nm[1024];
            snprintf(nm, sizeof nm, "%s", line + 7);
            strip_crlf(nm);
            for (int i = 0; i < s->nmat; ++i)
                if (strcmp(s->mats[i].name, nm) == 0) current_mat = i; /* last match wins, sticky if none */
            gmat[ngroups - 1] = current_mat;
        } else if (line[0] == 'v' && line[1] == ' ') {
            char *p = line + 2;
            float x = strtof(p, &p), y = strtof(p, &p), z = strtof(p, &p);
            fvec_push3(&vert, x, y, z);
        } else if (line[0] == 'v' && line[1] == 'n') {
            char *p = len >= 3 ? line + 3 : line + len;
            float x = strtof(p, &p), y = strtof(p, &p), z = strtof(p, &p);
            fvec_push3(&norm, x, y, z);
        } else if (line[0] == 'v' && line[1] == 't') { /* objimporter.hpp:222-228 */
            char *p = line + 2;
            float u = strtof(p, &p), v = strtof(p, &p);
            fvec_push3(&vt, u, v, 0.0f);
        } else if (line[0] == 'f' && line[1] == ' ') {
            /* every whitespace-separated element contributes its vertex id; no triangulation.  A non-empty
             * second field is a texture-coordinate id (:249-256); the third (normal id) is never used. */
            char *p = line + 2;
            for (;;) {
                while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
                if (!*p) break;
                int id = (int)strtol(p, NULL, 10);
                ivec_push(&groups[ngroups - 1], id - 1);
                char *q = p;
                while (*q && *q != ' ' && *q != '\t' && *q != '\r' && *q != '/') ++q;
                if (*q == '/' && q[1] && q[1] != '/' && q[1] != ' ' && q[1] != '\t' && q[1] != '\r')
                    ivec_push(&gtex[ngroups - 1], (int)strtol(q + 1, NULL, 10) - 1);
                while (*p && *p != ' ' && *p != '\t' && *p != '\r') ++p;
            }
        }
    }
    free(line);
    fclose(in);

    s->nv = (int)(vert.n / 3);
    s->raw_v = vert.p ? vert.p : (float *)calloc(3, sizeof(float));
    /* faces: consecutive index triples of each non-empty group in file order, mesh.hpp:454-481 */
    int nf = 0;
    for (int g = 0; g < ngroups; ++g) nf += (int)(groups[g].n / 3);
    s->nf = nf;
    s->fvid = (int *)calloc((size_t)nf * 3 + 1, sizeof(int));
    s->fmat = (int *)calloc((size_t)nf + 1, sizeof(int));
    /* Texture coordinates, objimporter.hpp:51-79,325-337 and mesh.hpp:463-468.  Per group a per-VERTEX table
     * (default (-1,-1)): the j-th texture id of the group goes to the vertex named by the j-th vertex id of
     * the group, later entries overwrite earlier ones.  Faces of the id-th NON-EMPTY group then read table
     * number id of ALL groups (the two numberings agree unless an empty group sits in front). */
    const int nvt = (int)(vt.n / 3);
    float **tables = NULL;
    if (nvt > 0) {
        s->ftex = (float *)calloc((size_t)nf * 6 + 1, sizeof(float));
        tables = (float **)calloc((size_t)ngroups, sizeof(float *));
        for (int g = 0; g < ngroups; ++g) {
            tables[g] = (float *)malloc((size_t)(s->nv > 0 ? s->nv : 1) * 2 * sizeof(float));
            for (int i = 0; i < 2 * s->nv; ++i) tables[g][i] = -1.0f;
            for (size_t j = 0; j < gtex[g].n && j < groups[g].n; ++j) {
                const int vid = groups[g].p[j], tid = gtex[g].p[j];
                if (vid < 0 || vid >= s->nv || tid < 0 || tid >= nvt) continue; /* out of range in the reference: undefined */
                tables[g][2 * vid] = vt.p[3 * tid], tables[g][2 * vid + 1] = vt.p[3 * tid + 1];
            }
        }
    }
    int f = 0, nonempty = 0;
    for (int g = 0; g < ngroups; ++g) {
        if (groups[g].n == 0) continue; /* objimporter.hpp:315-322: only non-empty groups become index buffers */
        for (size_t i = 0; i + 2 < groups[g].n; i += 3, ++f) {
            s->fvid[3 * f] = groups[g].p[i];
            s->fvid[3 * f + 1] = groups[g].p[i + 1];
            s->fvid[3 * f + 2] = groups[g].p[i + 2];
            s->fmat[f] = gmat[g];
            if (tables)
                for (int k = 0; k < 3; ++k) {
                    const int vid = s->fvid[3 * f + k];
                    const float *row = tables[nonempty];
                    s->ftex[6 * f + 2 * k] = vid >= 0 && vid < s->nv ? row[2 * vid] : -1.0f;
                    s->ftex[6 * f + 2 * k + 1] = vid >= 0 && vid < s->nv ? row[2 * vid + 1] : -1.0f;
                }
        }
        ++nonempty;
    }
    for (int g = 0; g < ngroups; ++g) {
        free(groups[g].p);
        free(gtex[g].p);
        if (tables) free(tables[g]);
    }
    free(tables);
    free(groups);
    free(gtex);
    free(gmat);
    free(vt.p);

    /* normals: file normals paired with vertices by position when the counts agree, else recomputed
     * (objimporter.hpp:290-300; the normal indices of the f records are never used) */
    s->raw_n = (float *)calloc((size_t)s->nv * 3 + 1, sizeof(float));
    if ((int)(norm.n / 3) == s->nv && s->nv > 0)
        memcpy(s->raw_n, norm.p, (size_t)s->nv * 3 * sizeof(float));
    else
        compute_vertex_normals(s);
    free(norm.p);
    finish_faces(s);

    s->mats_plain = (orc_material *)calloc((size_t)(s->nmat > 0 ? s->nmat : 1), sizeof(orc_material));
    for (int i = 0; i < s->nmat; ++i) s->mats_plain[i] = s->mats[i].m;
    load_material_textures(s);
    return s;
}

/* Textures for scenes given as arrays: texcoords = nf*6 or NULL; textures are appended one by one and tied
 * to materials by index. */
void orc_scene_set_texcoords(orc_scene *s, const float *texcoords) {
    free(s->ftex);
    s->ftex = NULL;
    if (!texcoords) return;
    s->ftex = (float *)malloc((size_t)s->nf * 6 * sizeof(float) + 4);
    memcpy(s->ftex, texcoords, (size_t)s->nf * 6 * sizeof(float));
}

int32_t orc_scene_add_texture(orc_scene *s, int32_t width, int32_t height, const float *rgb) {
    s->tex = (orc_tex *)realloc(s->tex, (size_t)(s->ntex + 1) * sizeof(orc_tex));
    orc_tex *t = &s->tex[s->ntex];
    memset(t, 0, sizeof *t);
    t->w = width, t->h = height;
    t->rgb = (float *)malloc((size_t)width * height * 3 * sizeof(float) + 4);
    memcpy(t->rgb, rgb, (size_t)width * height * 3 * sizeof(float));
    return s->ntex++;
}

void orc_scene_set_material_texture(orc_scene *s, int32_t material, int32_t texture) {
    if (!s->mat_tex) {
        s->mat_tex = (int *)malloc((size_t)(s->nmat > 0 ? s->nmat : 1) * sizeof(int));
        for (int i = 0; i < s->nmat; ++i) s->mat_tex[i] = -1;
    }
    if (material >= 0 && material < s->nmat) s->mat_tex[material] = texture;
}

const float *orc_face_texcoords(const orc_scene *s) { return s->ftex; }
int32_t orc_num_textures(const orc_scene *s) { return s->ntex; }
const float *orc_texture_data(const orc_scene *s, int32_t i, int32_t *width, int32_t *height) {
    if (i < 0 || i >= s->ntex) return NULL;
    *width = s->tex[i].w, *height = s->tex[i].h;
    return s->tex[i].rgb;
}
const int32_t *orc_material_textures(const orc_scene *s) { return s->mat_tex; }

orc_scene *orc_scene_from_arrays(const float *vertices, const float *normals, int32_t nv, const int32_t *faces,
                                 const int32_t *face_material, int32_t nf, const orc_material *materials,
                                 int32_t nmat) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->nv = nv, s->nf = nf, s->nmat = nmat;
    s->raw_v = (float *)malloc((size_t)nv * 3 * sizeof(float) + 4);
    memcpy(s->raw_v, vertices, (size_t)nv * 3 * sizeof(float));
    s->fvid = (int *)malloc((size_t)nf * 3 * sizeof(int) + 4);
    memcpy(s->fvid, faces, (size_t)nf * 3 * sizeof(int));
    s->fmat = (int *)malloc((size_t)nf * sizeof(int) + 4);
    for (int i = 0; i < nf; ++i) s->fmat[i] = face_material ? face_material[i] : -1;
    s->raw_n = (float *)calloc((size_t)nv * 3 + 1, sizeof(float));
    if (normals)
        memcpy(s->raw_n, normals, (size_t)nv * 3 * sizeof(float));
    else
        compute_vertex_normals(s);
    finish_faces(s);
    s->mats = (named_material *)calloc((size_t)(nmat > 0 ? nmat : 1), sizeof(named_material));
    s->mats_plain = (orc_material *)calloc((size_t)(nmat > 0 ? nmat : 1), sizeof(orc_material));
    for (int i = 0; i < nmat; ++i) s->mats[i].m = materials[i], s->mats_plain[i] = materials[i];
    return s;
}

static void free_node(orc_node *n) {
    if (!n) return;
    free_node(n->child[0]);
    free_node(n->child[1]);
    free(n->faces);
    free(n);
}

void orc_scene_free(orc_scene *s) {
    if (!s) return;
    free(s->raw_v), free(s->raw_n), free(s->fvid), free(s->fnormal), free(s->fmat);
    free(s->mats), free(s->mats_plain);
    free(s->ftex), free(s->mat_tex);
    for (int i = 0; i < s->ntex; ++i) free(s->tex[i].rgb);
    free(s->tex);
    free(s->P_v), free(s->P_n), free(s->P_D);
    free(s->light_pos), free(s->light_col), free(s->samples), free(s->spheres);
    free_node(s->root);
    free(s);
}

/* ------------------------------------ S3 + S4 ------------------------------------------------ */

void orc_scene_precompute(orc_scene *s, int32_t normalize_model) {
    /* Mesh::loadVertices, mesh.hpp:610-628 */
    float s_norm = 1.0f;
    v3 c = v3_make(0, 0, 0);
    if (s->nv > 0) {
        const float *v = s->raw_v;
        float xMax = v[0], xMin = v[0], yMax = v[1], yMin = v[1], zMax = v[2], zMin = v[2];
        for (int i = 0; i < s->nv; ++i) {
            xMax = f_max(v[3 * i], xMax), yMax = f_max(v[3 * i + 1], yMax), zMax = f_max(v[3 * i + 2], zMax);
            xMin = f_min(v[3 * i], xMin), yMin = f_min(v[3 * i + 1], yMin), zMin = f_min(v[3 * i + 2], zMin);
        }
        const float ext = f_max(f_max(fabsf(xMax - xMin), fabsf(yMax - yMin)), fabsf(zMax - zMin));
        s_norm = (float)(1.0 / (double)ext);
        c = v3_make((float)((double)(xMax + xMin) / 2.0), (float)((double)(yMax + yMin) / 2.0),
                    (float)((double)(zMax + zMin) / 2.0));
    }
    s->norm_scale = s_norm;
    s->center = c;
    if (normalize_model) { /* model.hpp:169-173 */
        model_matrix(s_norm, c, s->model);
    } else {
        model_matrix(1.0f, v3_make(0, 0, 0), s->model);
        s->model[3] = s->model[7] = s->model[11] = 0.0f;
    }
    /* Flyscene::precomputeData, flyscene.cpp:165-219 */
    free(s->P_v), free(s->P_n), free(s->P_D);
    s->P_v = (float *)calloc((size_t)s->nv * 3 + 1, sizeof(float));
    s->P_n = (float *)calloc((size_t)s->nv * 3 + 1, sizeof(float));
    s->P_D = (float *)calloc((size_t)s->nf + 1, sizeof(float));
    for (int f = 0; f < s->nf; ++f) {
        v3 v0t = v3_make(0, 0, 0);
        for (int k = 0; k < 3; ++k) {
            const int id = s->fvid[3 * f + k];
            v3 p = model_point(s->model, v3_make(s->raw_v[3 * id], s->raw_v[3 * id + 1], s->raw_v[3 * id + 2]));
            s->P_v[3 * id] = p.x, s->P_v[3 * id + 1] = p.y, s->P_v[3 * id + 2] = p.z;
            v3 n = v3_normalized(v3_make(s->raw_n[3 * id], s->raw_n[3 * id + 1], s->raw_n[3 * id + 2]));
            s->P_n[3 * id] = n.x, s->P_n[3 * id + 1] = n.y, s->P_n[3 * id + 2] = n.z;
            if (k == 0) v0t = p;
        }
        v3 fn = v3_make(s->fnormal[3 * f], s->fnormal[3 * f + 1], s->fnormal[3 * f + 2]);
        s->P_D[f] = v3_dot(fn, v0t); /* :209 */
    }
}

/* ------------------------------------ S5: light samples -------------------------------------- */

void orc_scene_set_lights(orc_scene *s, const float *pos_color, int32_t nlights, int32_t hard_shadow,
                          int32_t soft_points, float soft_radius) {
    free(s->light_pos), free(s->light_col), free(s->samples);
    s->nl = s->nl_shade = nlights;
    s->spl = (hard_shadow ? 1 : 0) + soft_points;
    s->light_pos = (float *)calloc((size_t)nlights * 3 + 1, sizeof(float));
    s->light_col = (float *)calloc((size_t)nlights * 3 + 1, sizeof(float));
    s->samples = (float *)calloc((size_t)nlights * (size_t)s->spl * 3 + 1, sizeof(float));
    for (int i = 0; i < nlights; ++i) {
        memcpy(s->light_pos + 3 * i, pos_color + 6 * i, 12);
        memcpy(s->light_col + 3 * i, pos_color + 6 * i + 3, 12);
        float *out = s->samples + (size_t)i * s->spl * 3;
        int k = 0;
        if (hard_shadow) { /* flyscene.cpp:228-233 */
            memcpy(out, pos_color + 6 * i, 12);
            k = 1;
        }
        for (int n = 0; n < soft_points; ++n, ++k) { /* flyscene.cpp:240-247 */
            const float theta = (2.0f * (float)n * (float)M_PI) / soft_points;
            const float phi = acosf(1 - 2 * ((float)n / (soft_points * (float)M_PI)));
            const float x = soft_radius * sinf(phi) * cosf(theta);
            const float y = soft_radius * sinf(phi) * sinf(theta);
            const float z = soft_radius * cosf(phi);
            out[3 * k] = pos_color[6 * i] + x;
            out[3 * k + 1] = pos_color[6 * i + 1] + y;
            out[3 * k + 2] = pos_color[6 * i + 2] + z;
        }
    }
}

/* Flyscene::addLight, flyscene.hpp:76: the light is appended to `lights` only.  precomputeLights runs once, in
 * initialize (flyscene.cpp:117), so the new light gets no shadow sample points: shadeOffFace's loop (:505) sees it,
 * getLightIntensity's loop over precomputedData.lights (:880) does not. */
void orc_scene_add_light(orc_scene *s, const float *pos_color) {
    const int n = s->nl_shade + 1;
    s->light_pos = (float *)realloc(s->light_pos, ((size_t)n * 3 + 1) * sizeof(float));
    s->light_col = (float *)realloc(s->light_col, ((size_t)n * 3 + 1) * sizeof(float));
    memcpy(s->light_pos + 3 * (size_t)s->nl_shade, pos_color, 12);
    memcpy(s->light_col + 3 * (size_t)s->nl_shade, pos_color + 3, 12);
    s->nl_shade = n;
}

void orc_scene_set_spheres(orc_scene *s, const float *spheres, int32_t n) {
    free(s->spheres);
    s->ns = n;
    s->spheres = (float *)calloc((size_t)n * 4 + 1, sizeof(float));
    if (n > 0) memcpy(s->spheres, spheres, (size_t)n * 16);
}

/* ------------------------------------ B1 + B2: BVH -------------------------------------------- */

static inline v3 pv(const orc_scene *s, int id) { return v3_make(s->P_v[3 * id], s->P_v[3 * id + 1], s->P_v[3 * id + 2]); }

/* boundingBox::splitBox, boundingBox.h:69-194 */
static void split_box(orc_scene *s, orc_node *b) {
    ++s->node_count;
    const int base_case = 10; /* boundingBox.h:34 */
    if (b->nfaces < base_case) {
        ++s->leaf_count;
        return;
    }
    const float side[3] = {b->vmax.x - b->vmin.x, b->vmax.y - b->vmin.y, b->vmax.z - b->vmin.z};
    int axis = 0; /* std::max_element: first of the largest */
    if (side[1] > side[axis]) axis = 1;
    if (side[2] > side[axis]) axis = 2;

    float sum = 0, weight = 0;
    for (int i = 0; i < b->nfaces; ++i) {
        const int *id = s->fvid + 3 * b->faces[i];
        sum += s->P_v[3 * id[0] + axis] + s->P_v[3 * id[1] + axis] + s->P_v[3 * id[2] + axis];
        weight += 3;
    }
    const float avg = sum / weight;

    orc_node *lower = (orc_node *)calloc(1, sizeof(orc_node));
    orc_node *upper = (orc_node *)calloc(1, sizeof(orc_node));
    lower->faces = (int *)malloc((size_t)b->nfaces * sizeof(int));
    upper->faces = (int *)malloc((size_t)b->nfaces * sizeof(int));
    v3 upperMin = b->vmax, upperMax = b->vmin, lowerMin = b->vmax, lowerMax = b->vmin;
    unsigned amountInRandomBox = 0;
    for (int i = 0; i < b->nfaces; ++i) {
        const int it = b->faces[i];
        const int *id = s->fvid + 3 * it;
        const v3 v0 = pv(s, id[0]), v1 = pv(s, id[1]), v2 = pv(s, id[2]);
        const float first = (&v0.x)[axis], sec = (&v1.x)[axis], third = (&v2.x)[axis];
        const int isInUpperBox = first > avg && sec > avg && third > avg;
        const int isInLowerBox = first < avg && sec < avg && third < avg;
        if (isInUpperBox || (!isInLowerBox && amountInRandomBox % 2 == 0)) {
            upper->faces[upper->nfaces++] = it;
            upperMin = v3_make(f_min(v0.x, f_min(v1.x, f_min(v2.x, upperMin.x))), f_min(v0.y, f_min(v1.y, f_min(v2.y, upperMin.y))),
                               f_min(v0.z, f_min(v1.z, f_min(v2.z, upperMin.z))));
            upperMax = v3_make(f_max(v0.x, f_max(v1.x, f_max(v2.x, upperMax.x))), f_max(v0.y, f_max(v1.y, f_max(v2.y, upperMax.y))),
                               f_max(v0.z, f_max(v1.z, f_max(v2.z, upperMax.z))));
            if (!isInUpperBox) amountInRandomBox++;
        } else {
            lower->faces[lower->nfaces++] = it;
            lowerMin = v3_make(f_min(v0.x, f_min(v1.x, f_min(v2.x, lowerMin.x))), f_min(v0.y, f_min(v1.y, f_min(v2.y, lowerMin.y))),
                               f_min(v0.z, f_min(v1.z, f_min(v2.z, lowerMin.z))));
            lowerMax = v3_make(f_max(v0.x, f_max(v1.x, f_max(v2.x, lowerMax.x))), f_max(v0.y, f_max(v1.y, f_max(v2.y, lowerMax.y))),
                               f_max(v0.z, f_max(v1.z, f_max(v2.z, lowerMax.z))));
            if (!isInLowerBox) amountInRandomBox++;
        }
    }
    upper->vmin = upperMin, upper->vmax = upperMax;
    lower->vmin = lowerMin, lower->vmax = lowerMax;

    if ((double)lower->nfaces < 0.8 * (double)b->nfaces && (double)upper->nfaces < 0.8 * (double)b->nfaces) {
        split_box(s, lower);
        split_box(s, upper);
        b->child[0] = lower;
        b->child[1] = upper;
    } else {
        ++s->leaf_count;
        free_node(lower);
        free_node(upper);
    }
}

static int node_depth(const orc_node *n, int d) {
    if (!n->child[0]) return d;
    const int a = node_depth(n->child[0], d + 1), b = node_depth(n->child[1], d + 1);
    return a > b ? a : b;
}

void orc_scene_build_bvh(orc_scene *s) {
    free_node(s->root);
    s->node_count = s->leaf_count = 0;
    orc_node *root = (orc_node *)calloc(1, sizeof(orc_node));
    /* Flyscene::boundingVectors, flyscene.cpp:841-868: over all nv entries of the vertex array */
    v3 vmin = v3_make(0, 0, 0), vmax = vmin;
    if (s->nv > 0) {
        vmin = vmax = pv(s, 0);
        for (int i = 0; i < s->nv; ++i) {
            const v3 p = pv(s, i);
            vmin = v3_make(f_min(vmin.x, p.x), f_min(vmin.y, p.y), f_min(vmin.z, p.z));
            vmax = v3_make(f_max(vmax.x, p.x), f_max(vmax.y, p.y), f_max(vmax.z, p.z));
        }
    }
    root->vmin = vmin, root->vmax = vmax;
    root->faces = (int *)malloc((size_t)(s->nf > 0 ? s->nf : 1) * sizeof(int));
    root->nfaces = s->nf;
    for (int i = 0; i < s->nf; ++i) root->faces[i] = i; /* flyscene.cpp:129-132 */
    s->root = root;
    split_box(s, root);
    s->depth = node_depth(root, 0);
}

/* ------------------------------------ accessors ------------------------------------------------ */

int32_t orc_num_vertices(const orc_scene *s) { return s->nv; }
int32_t orc_num_faces(const orc_scene *s) { return s->nf; }
int32_t orc_num_materials(const orc_scene *s) { return s->nmat; }
int32_t orc_num_lights(const orc_scene *s) { return s->nl; }
int32_t orc_num_shading_lights(const orc_scene *s) { return s->nl_shade; }
int32_t orc_samples_per_light(const orc_scene *s) { return s->spl; }
const float *orc_raw_vertices(const orc_scene *s) { return s->raw_v; }
const float *orc_raw_normals(const orc_scene *s) { return s->raw_n; }
const float *orc_vertices(const orc_scene *s) { return s->P_v; }
const float *orc_normals(const orc_scene *s) { return s->P_n; }
const int32_t *orc_face_vertex_ids(const orc_scene *s) { return s->fvid; }
const float *orc_face_normals(const orc_scene *s) { return s->fnormal; }
const float *orc_face_origin_distance(const orc_scene *s) { return s->P_D; }
const int32_t *orc_face_material_ids(const orc_scene *s) { return s->fmat; }
const orc_material *orc_materials(const orc_scene *s) { return s->mats_plain; }
const float *orc_light_samples(const orc_scene *s) { return s->samples; }
const float *orc_model_matrix(const orc_scene *s) { return s->model; }
int32_t orc_bvh_node_count(const orc_scene *s) { return s->node_count; }
int32_t orc_bvh_leaf_count(const orc_scene *s) { return s->leaf_count; }
int32_t orc_bvh_depth(const orc_scene *s) { return s->depth; }

static void face_order_rec(const orc_node *n, int32_t *out, int *pos) {
    if (!n->child[0]) {
        for (int i = 0; i < n->nfaces; ++i) out[n->faces[i]] = (*pos)++;
        return;
    }
    face_order_rec(n->child[0], out, pos);
    face_order_rec(n->child[1], out, pos);
}
void orc_bvh_face_order(const orc_scene *s, int32_t *out) {
    int pos = 0;
    if (s->root) face_order_rec(s->root, out, &pos);
}
static void dump_rec(const orc_node *n, float *boxes, int32_t *info, int *k) {
    const int i = (*k)++;
    boxes[6 * i] = n->vmin.x, boxes[6 * i + 1] = n->vmin.y, boxes[6 * i + 2] = n->vmin.z;
    boxes[6 * i + 3] = n->vmax.x, boxes[6 * i + 4] = n->vmax.y, boxes[6 * i + 5] = n->vmax.z;
    info[2 * i] = n->child[0] ? 0 : 1;
    info[2 * i + 1] = n->nfaces;
    if (n->child[0]) {
        dump_rec(n->child[0], boxes, info, k);
        dump_rec(n->child[1], boxes, info, k);
    }
}
void orc_bvh_dump(const orc_scene *s, float *boxes, int32_t *info) {
    int k = 0;
    if (s->root) dump_rec(s->root, boxes, info, &k);
}

void orc_camera_default(orc_camera *c, int32_t width, int32_t height) {
    /* Flycamera::reset, flycamera.hpp:76-86,166-191: view = Translate(0,0,-2); flyscene.cpp:35-36 */
    memset(c, 0, sizeof(*c));
    c->origin[0] = 0, c->origin[1] = 0, c->origin[2] = 2;
    c->inv_view[0] = c->inv_view[5] = c->inv_view[10] = 1.0f;
    c->inv_view[11] = 2.0f;
    c->viewport[2] = (float)width, c->viewport[3] = (float)height;
    c->aspect_ratio = (float)width / (float)height;
    c->fovy = 60.0f;
}

void orc_params_default(orc_params *p, int32_t width, int32_t height) {
    memset(p, 0, sizeof(*p));
    p->width = width, p->height = height;
    p->ssaa_level = 1;
    p->max_reflections = 5;
    p->reflect_threshold = 0.25f;
    p->bounce_offset = 0.01f;
    p->t_epsilon = 0.00001f;
    p->background[0] = p->background[1] = p->background[2] = (float)0.4;
    p->num_threads = 0;
    p->row_offset = 0, p->row_step = 1;
}

/* ------------------------------------ tracing -------------------------------------------------- */

typedef struct {
    const orc_node **p;
    size_t n, cap;
} leafvec;

typedef struct {
    orc_scene *s;
    const orc_camera *cam;
    const orc_params *prm;
    leafvec leaves; /* thread_local intersectingFaces, flyscene.cpp:795 */
    orc_counters cnt;
    float *rgb, *t_hit, *light;
    int32_t *face_id;
    unsigned tid, nthreads;
    int32_t *nan_mask;
    int nan_flag; /* set when a NaN plane distance or a 0 / 0 in a box test is met; reset per pixel */
} tctx;

/* boundingBox::boxIntersection, boundingBox.h:257-286.
 * A ray that runs IN the plane of a box side (direction component exactly zero, origin coordinate equal to the side's)
 * makes the reference divide 0 by 0; with its min/max and comparisons a box that ENDS at that plane is then turned away
 * and one that STARTS there is taken, so the outcome depends on the boxes the hierarchy happens to have.  Such a ray is
 * flagged like a NaN plane distance (nan_flag), and with reject_nan set the slab counts as passed, which is what exact
 * arithmetic gives (the ray touches the box) and what the HIP path, which has other boxes, defines. */
static int box_intersection(tctx *c, const orc_node *b, v3 o, v3 d) {
    float tx_min = (b->vmin.x - o.x) / d.x, tx_max = (b->vmax.x - o.x) / d.x;
    float ty_min = (b->vmin.y - o.y) / d.y, ty_max = (b->vmax.y - o.y) / d.y;
    float tz_min = (b->vmin.z - o.z) / d.z, tz_max = (b->vmax.z - o.z) / d.z;
    if (tx_min != tx_min || tx_max != tx_max || ty_min != ty_min || ty_max != ty_max || tz_min != tz_min || tz_max != tz_max) {
        c->nan_flag = 1;
        if (c->prm->reject_nan) {
            if (tx_min != tx_min || tx_max != tx_max) tx_min = -INFINITY, tx_max = INFINITY;
            if (ty_min != ty_min || ty_max != ty_max) ty_min = -INFINITY, ty_max = INFINITY;
            if (tz_min != tz_min || tz_max != tz_max) tz_min = -INFINITY, tz_max = INFINITY;
        }
    }
    if (tx_min > tx_max) {
        float t = tx_min;
        tx_min = tx_max, tx_max = t;
    }
    if (ty_min > ty_max) {
        float t = ty_min;
        ty_min = ty_max, ty_max = t;
    }
    if (tz_min > tz_max) {
        float t = tz_min;
        tz_min = tz_max, tz_max = t;
    }
    const float tin_x = f_min(tx_min, tx_max), tout_x = f_max(tx_min, tx_max);
    const float tin_y = f_min(ty_min, ty_max), tout_y = f_max(ty_min, ty_max);
    const float tin_z = f_min(tz_min, tz_max), tout_z = f_max(tz_min, tz_max);
    const float tin = f_max(tin_x, f_max(tin_y, tin_z));
    const float tout = f_min(tout_x, f_min(tout_y, tout_z));
    return !(tin > tout || tout < 0);
}

/* boundingBox::intersectingBoxes, boundingBox.h:295-307 */
static void intersecting_boxes(tctx *c, const orc_node *b, v3 o, v3 d) {
    ++c->cnt.box_tests;
    if (!box_intersection(c, b, o, d) && !c->prm->all_faces) return;
    if (!b->child[0]) {
        if (c->leaves.n == c->leaves.cap) {
            c->leaves.cap = c->leaves.cap ? c->leaves.cap * 2 : 256;
            c->leaves.p = (const orc_node **)realloc((void *)c->leaves.p, c->leaves.cap * sizeof(void *));
        }
        c->leaves.p[c->leaves.n++] = b;
        ++c->cnt.leaves_visited;
    } else {
        intersecting_boxes(c, b->child[0], o, d);
        intersecting_boxes(c, b->child[1], o, d);
    }
}

typedef struct {
    int face_id;
    v3 hitpoint, reflection;
    float t;
} hit_t;

/* Flyscene::sphereIntersection, flyscene.cpp:681-725 */
static int sphere_intersection(tctx *c, float *max_depth, v3 o, v3 d, hit_t *h) {
    int has = 0;
    const orc_scene *s = c->s;
    for (int i = 0; i < s->ns; ++i) {
        const v3 center = v3_make(s->spheres[4 * i], s->spheres[4 * i + 1], s->spheres[4 * i + 2]);
        const float radius = s->spheres[4 * i + 3];
        const v3 oc = v3_sub(center, o);
        const float proj = v3_dot(oc, d);
        /* pow(float, int) promotes to double: :691-692 */
        const float centerToRay2 = (float)((double)v3_sqnorm(oc) - (double)proj * (double)proj);
        const float radius2 = (float)((double)radius * (double)radius);
        if (centerToRay2 > radius2) continue;
        const float inToProjection = sqrtf(radius2 - centerToRay2);
        float inT = proj - inToProjection, outT = proj + inToProjection;
        if (inT > outT) {
            float t = inT;
            inT = outT, outT = t;
        }
        if (inT < 0 && outT < 0)
            continue;
        else if (inT < 0)
            inT = outT;
        if (inT < c->prm->t_epsilon || inT > *max_depth) continue;
        const v3 hp = v3_add(o, v3_scale(inT, d));
        has = 1;
        *max_depth = inT;
        const v3 normal = v3_normalized(v3_sub(hp, center));
        h->reflection = v3_sub(d, v3_scale(2 * v3_dot(d, normal), normal));
        h->hitpoint = hp;
        h->face_id = -1;
        h->t = inT;
    }
    return has;
}

/* Flyscene::triangleIntersection, flyscene.cpp:727-793 */
static int triangle_intersection(tctx *c, float *max_depth, v3 o, v3 d, hit_t *h) {
    const orc_scene *s = c->s;
    c->leaves.n = 0;
    if (s->root) intersecting_boxes(c, s->root, o, d);
    int has = 0;
    const float eps = c->prm->t_epsilon;
    for (size_t l = 0; l < c->leaves.n; ++l) {
        const orc_node *leaf = c->leaves.p[l];
        for (int k = 0; k < leaf->nfaces; ++k) {
            const int i = leaf->faces[k];
            const v3 normal = v3_make(s->fnormal[3 * i], s->fnormal[3 * i + 1], s->fnormal[3 * i + 2]);
            const float originDistance = s->P_D[i];
            ++c->cnt.tri_plane;
            const float tHit = (originDistance - v3_dot(o, normal)) / v3_dot(d, normal);
            if (tHit < eps || tHit > *max_depth) continue;
            if (tHit != tHit) {
                c->nan_flag = 1;
                if (c->prm->reject_nan) continue;
            }
            ++c->cnt.tri_edge;
            const int *id = s->fvid + 3 * i;
            const v3 v0 = pv(s, id[0]), v1 = pv(s, id[1]), v2 = pv(s, id[2]);
            const v3 hp = v3_add(o, v3_scale(tHit, d));
            const float a = v3_dot(v3_cross(v3_sub(v1, v0), v3_sub(hp, v0)), normal);
            if (a < 0) continue;
            const float b = v3_dot(v3_cross(v3_sub(v2, v1), v3_sub(hp, v1)), normal);
            if (b < 0) continue;
            const float cc = v3_dot(v3_cross(v3_sub(v0, v2), v3_sub(hp, v2)), normal);
            if (cc < 0) continue;
            if (tHit != tHit) ++c->cnt.nan_accepts;
            has = 1;
            *max_depth = tHit;
            h->reflection = v3_sub(d, v3_scale(2 * v3_dot(d, normal), normal));
            h->hitpoint = hp;
            h->face_id = i;
            h->t = tHit;
        }
    }
    return has;
}

/* Flyscene::intersects, flyscene.cpp:669-679 */
static int intersects(tctx *c, v3 o, v3 d, hit_t *h) {
    float max_depth = FLT_MAX;
    const int a = sphere_intersection(c, &max_depth, o, d, h);
    const int b = triangle_intersection(c, &max_depth, o, d, h);
    return a || b;
}

/* Flyscene::getLightIntensity, flyscene.cpp:870-896.  The sample position is passed as the ray
 * DIRECTION (:884), unnormalised, from the un-offset hit position. */
static float light_intensity(tctx *c, v3 hit) {
    const orc_scene *s = c->s;
    float li = 0;
    hit_t h;
    for (int l = 0; l < s->nl; ++l) {
        float reaching = 0;
        for (int k = 0; k < s->spl; ++k) {
            const float *q = s->samples + ((size_t)l * s->spl + k) * 3;
            ++c->cnt.rays_shadow;
            if (!intersects(c, hit, v3_make(q[0], q[1], q[2]), &h)) reaching++;
        }
        li += reaching / (float)s->spl;
    }
    return f_min(li, 1.0f);
}

/* Flyscene::shadeOffFace, flyscene.cpp:461-579 */
static v3 shade_off_face(tctx *c, int face, v3 origin, v3 hit, v3 LI) {
    const orc_scene *s = c->s;
    if (face == -1) return v3_make(0.7f, 0.9f, 0.0f);
    const int mat = s->fmat[face];
    if (mat == -1) return v3_make(0.5f, 0.5f, 0.5f);
    const orc_material *m = &s->mats_plain[mat];
    const int *id = s->fvid + 3 * face;
    const v3 v0 = pv(s, id[0]), v1 = pv(s, id[1]), v2 = pv(s, id[2]);
    const v3 n0 = v3_make(s->P_n[3 * id[0]], s->P_n[3 * id[0] + 1], s->P_n[3 * id[0] + 2]);
    const v3 n1 = v3_make(s->P_n[3 * id[1]], s->P_n[3 * id[1] + 1], s->P_n[3 * id[1] + 2]);
    const v3 n2 = v3_make(s->P_n[3 * id[2]], s->P_n[3 * id[2] + 1], s->P_n[3 * id[2] + 2]);
    /* `norm() * 0.5` is a double that Eigen converts back to float before scaling (exact halving) */
    const float a12 = v3_norm(v3_cross(v3_sub(v1, hit), v3_sub(v2, hit))) * 0.5f;
    const float a02 = v3_norm(v3_cross(v3_sub(v0, hit), v3_sub(v2, hit))) * 0.5f;
    const float a01 = v3_norm(v3_cross(v3_sub(v0, hit), v3_sub(v1, hit))) * 0.5f;
    const v3 N = v3_normalized(v3_add(v3_add(v3_scale(a12, n0), v3_scale(a02, n1)), v3_scale(a01, n2)));
    const v3 ka = v3_make(m->ka[0], m->ka[1], m->ka[2]);
    const v3 kd = v3_make(m->kd[0], m->kd[1], m->kd[2]);
    const v3 ks = v3_make(m->ks[0], m->ks[1], m->ks[2]);
    /* :476-481: the face carries three texture coordinates (any `vt` in the file gives every face three) and
     * the material names a texture */
    const orc_tex *tex = s->ftex && s->mat_tex && s->mat_tex[mat] >= 0 ? &s->tex[s->mat_tex[mat]] : NULL;
    v3 texel = v3_make(0, 0, 0);
    if (tex) {
        /* :519-539.  The weights are the absolute sub-triangle areas, NOT divided by the face area.  uacc and
         * vacc are floors of fractional parts, i.e. always 0, so the "interpolation" is 1*cij + 0*... = the
         * texel at (projU, projV).  The reference indexes the vector unchecked; an index outside the image
         * (undefined there) is clamped to it here. */
        const float *tc = s->ftex + 6 * face;
        const float u = (a12 * tc[0] + a02 * tc[2]) + a01 * tc[4];
        const float v = (a12 * tc[1] + a02 * tc[3]) + a01 * tc[5];
        const float fu = floorf(u * (float)tex->w), fv = floorf(v * (float)tex->h);
        const long last = (long)tex->w * tex->h - 1;
        double idx = (double)fv * (double)tex->w + (double)fu;
        long at = !(idx >= 0) ? 0 : idx > (double)last ? last : (long)idx;
        if (last < 0) {
            tex = NULL; /* unreadable file: the reference reads an empty vector; here the material colour is used */
        } else {
            texel = v3_make(tex->rgb[3 * at], tex->rgb[3 * at + 1], tex->rgb[3 * at + 2]);
        }
    }
    v3 color = v3_make(0, 0, 0);
    for (int l = 0; l < s->nl_shade; ++l) { /* :505, every entry of `lights` */
        const v3 lp = v3_make(s->light_pos[3 * l], s->light_pos[3 * l + 1], s->light_pos[3 * l + 2]);
        const v3 L = v3_normalized(v3_sub(lp, hit));
        const v3 ambient = v3_mul(LI, ka);
        const float cos1 = fmaxf(0, v3_dot(L, N));
        const v3 diffuse = v3_scale(cos1, v3_mul(LI, tex ? texel : kd)); /* :514 / :541 */
        const v3 E = v3_normalized(v3_sub(origin, hit));
        const v3 R = v3_normalized(v3_add(v3_neg(L), v3_scale(2.f * v3_dot(L, N), N)));
        const float cos2 = (float)fmax(0, (double)v3_dot(R, E));
        const v3 specular = v3_scale(powf(cos2, m->shininess), v3_mul(LI, ks));
        const v3 sum = v3_add(v3_add(diffuse, specular), ambient);
        color = v3_add(color, v3_clamp01(sum));
    }
    return color;
}

/* Flyscene::traceRay, flyscene.cpp:581-667 */
static v3 trace_ray(tctx *c, v3 origin, v3 direction, int depth, hit_t *primary, float *primary_li, int *primary_hit) {
    hit_t h;
    h.face_id = -2;
    if (depth == 0)
        ++c->cnt.rays_primary;
    else
        ++c->cnt.rays_bounce;
    const int b = intersects(c, origin, direction, &h);
    if (depth == 0) {
        *primary_hit = b;
        if (b) *primary = h;
    }
    if (b) {
        const float li = light_intensity(c, h.hitpoint);
        if (depth == 0) *primary_li = li;
        const v3 LI = v3_make(li, li, li);
        const v3 local = shade_off_face(c, h.face_id, origin, h.hitpoint, LI);
        if (depth < c->prm->max_reflections) {
            if (h.face_id == -1) return local;
            const int mat = c->s->fmat[h.face_id];
            if (mat == -1) return local;
            const orc_material *m = &c->s->mats_plain[mat];
            const float E = c->prm->reflect_threshold;
            if (m->ks[0] > E || m->ks[1] > E || m->ks[2] > E) {
                const v3 r = v3_normalized(h.reflection);
                const v3 refl = trace_ray(c, v3_add(h.hitpoint, v3_scale(c->prm->bounce_offset, r)), r, depth + 1, NULL, NULL, NULL);
                const v3 weighted = v3_mul(v3_mul(refl, v3_make(m->ks[0], m->ks[1], m->ks[2])), LI);
                return v3_add(local, weighted);
            }
        }
        return local;
    }
    if (depth == 0) return v3_make(c->prm->background[0], c->prm->background[1], c->prm->background[2]);
    return v3_make(c->prm->background_deep[0], c->prm->background_deep[1], c->prm->background_deep[2]);
}

/* Camera::screenToWorld, camera.hpp:158-176,271-274 */
static v3 screen_point(const orc_camera *cam, float px, float py) {
    const float *vp = cam->viewport;
    v3 n;
    n.x = (float)(2.0 * (double)(px - vp[0]) / (double)vp[2] - 1.0);
    n.y = (float)(1.0 - 2.0 * (double)(py - vp[1]) / (double)vp[3]);
    n.z = (float)-1.0;
    const float persp = (float)((double)1.0f / tan((double)(cam->fovy / 2.0f) * (M_PI / (double)180.0f)));
    const float scale = (float)(1.0 / (double)persp);
    n.x *= cam->aspect_ratio * scale;
    n.y *= scale;
    return affine_point(cam->inv_view, n);
}

/* ... + the normalisation in tracePixels, flyscene.cpp:820-822 */
static v3 screen_ray(const orc_camera *cam, float px, float py) {
    const v3 world = screen_point(cam, px, py);
    const v3 origin = v3_make(cam->origin[0], cam->origin[1], cam->origin[2]);
    return v3_normalized(v3_sub(world, origin));
}

void orc_screen_points(const orc_camera *cam, int32_t width, int32_t height, int32_t ystep, float *out) {
    if (ystep < 1) ystep = 1;
    for (int y = 0; y < height; y += ystep)
        for (int x = 0; x < width; ++x) {
            const v3 p = screen_point(cam, (float)x, (float)y);
            *out++ = p.x, *out++ = p.y, *out++ = p.z;
        }
}

void orc_screen_ray(const orc_camera *cam, float px, float py, float *direction) {
    const v3 d = screen_ray(cam, px, py);
    direction[0] = d.x, direction[1] = d.y, direction[2] = d.z;
}

/* Flyscene::tracePixels, flyscene.cpp:797-839 */
static void *trace_pixels(void *arg) {
    tctx *c = (tctx *)arg;
    const orc_params *p = c->prm;
    const int W = p->width, H = p->height, S = p->ssaa_level;
    const v3 origin = v3_make(c->cam->origin[0], c->cam->origin[1], c->cam->origin[2]);
    const int step = p->row_step > 0 ? p->row_step : 1;
    for (int x = 0; x < W; ++x) {
        for (int yi = (int)c->tid;; yi += (int)c->nthreads) {
            const long yl = (long)p->row_offset + (long)yi * step;
            if (yl >= H) break;
            const int y = (int)yl;
            v3 color = v3_make(0, 0, 0);
            hit_t ph;
            float pli = 0;
            int phit = 0;
            c->nan_flag = 0;
            for (int xs = 0; xs < S; ++xs)
                for (int ys = 0; ys < S; ++ys) {
                    const float cx = (float)x + (float)xs / S;
                    const float cy = (float)y + (float)ys / S;
                    const v3 d = screen_ray(c->cam, cx, cy);
                    pli = 0;
                    color = v3_add(color, trace_ray(c, origin, d, 0, &ph, &pli, &phit));
                }
            const size_t idx = (size_t)y * W + x;
            if (c->rgb) {
                const v3 px = v3_divs(color, (float)(S * S));
                c->rgb[3 * idx] = px.x, c->rgb[3 * idx + 1] = px.y, c->rgb[3 * idx + 2] = px.z;
            }
            if (c->face_id) c->face_id[idx] = phit ? ph.face_id : -2;
            if (c->t_hit) c->t_hit[idx] = phit ? ph.t : 0.0f;
            if (c->light) c->light[idx] = phit ? pli : 0.0f;
            if (c->nan_mask) c->nan_mask[idx] = c->nan_flag;
        }
    }
    return NULL;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Flyscene::raytraceScene, flyscene.cpp:381-459 (PPM writing is a separate call) */
int32_t orc_render(orc_scene *s, const orc_camera *cam, const orc_params *p, float *rgb, int32_t *face_id, float *t_hit,
                   float *light, int32_t *nan_mask, orc_counters *counters) {
    if (!s || !cam || !p || p->width <= 0 || p->height <= 0 || p->ssaa_level < 1) return 1;
    int nt = p->num_threads;
    if (nt <= 0) nt = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (nt < 1) nt = 1;
    tctx *ctx = (tctx *)calloc((size_t)nt, sizeof(tctx));
    pthread_t *th = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
    const double t0 = now_s();
    for (int i = 0; i < nt; ++i) {
        ctx[i].s = s, ctx[i].cam = cam, ctx[i].prm = p;
        ctx[i].rgb = rgb, ctx[i].face_id = face_id, ctx[i].t_hit = t_hit, ctx[i].light = light, ctx[i].nan_mask = nan_mask;
        ctx[i].tid = (unsigned)i, ctx[i].nthreads = (unsigned)nt;
        pthread_create(&th[i], NULL, trace_pixels, &ctx[i]);
    }
    for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
    const double t1 = now_s();
    if (counters) {
        memset(counters, 0, sizeof(*counters));
        for (int i = 0; i < nt; ++i) {
            counters->rays_primary += ctx[i].cnt.rays_primary;
            counters->rays_shadow += ctx[i].cnt.rays_shadow;
            counters->rays_bounce += ctx[i].cnt.rays_bounce;
            counters->box_tests += ctx[i].cnt.box_tests;
            counters->leaves_visited += ctx[i].cnt.leaves_visited;
            counters->tri_plane += ctx[i].cnt.tri_plane;
            counters->tri_edge += ctx[i].cnt.tri_edge;
            counters->nan_accepts += ctx[i].cnt.nan_accepts;
        }
        counters->seconds = t1 - t0;
    }
    for (int i = 0; i < nt; ++i) free((void *)ctx[i].leaves.p);
    free(ctx);
    free(th);
    return 0;
}

int32_t orc_intersects(orc_scene *s, const float *origin, const float *direction, const orc_params *p, int32_t *face_id,
                       float *t_hit, float *hitpoint, float *reflection) {
    tctx c;
    memset(&c, 0, sizeof c);
    c.s = s, c.prm = p;
    hit_t h;
    h.face_id = -2, h.t = 0;
    h.hitpoint = h.reflection = v3_make(0, 0, 0);
    const int b = intersects(&c, v3_make(origin[0], origin[1], origin[2]), v3_make(direction[0], direction[1], direction[2]), &h);
    free((void *)c.leaves.p);
    if (face_id) *face_id = b ? h.face_id : -2;
    if (t_hit) *t_hit = b ? h.t : 0.0f;
    if (hitpoint && b) hitpoint[0] = h.hitpoint.x, hitpoint[1] = h.hitpoint.y, hitpoint[2] = h.hitpoint.z;
    if (reflection && b) reflection[0] = h.reflection.x, reflection[1] = h.reflection.y, reflection[2] = h.reflection.z;
    return b;
}

/* writePPMImage, ppmIO.hpp:188-209 */
int32_t orc_write_ppm(const char *path, const float *rgb, int32_t width, int32_t height) {
    FILE *out = fopen(path, "w");
    if (!out) return 1;
    fprintf(out, "P3\n%d %d\n255\n", width, height);
    for (int j = 0; j < height; ++j) {
        for (int i = 0; i < width; ++i) {
            const float *p = rgb + 3 * ((size_t)j * width + i);
            int c[3];
            for (int k = 0; k < 3; ++k) {
                c[k] = (int)(255 * p[k]);
                if (c[k] > 255) c[k] = 255;
            }
            fprintf(out, "%d %d %d ", c[0], c[1], c[2]);
        }
        fprintf(out, "\n");
    }
    fclose(out);
    return 0;
}
