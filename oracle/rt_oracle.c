/*
 * rt_oracle.c - CPU restatement of the reference's per-pixel render loop.
 * TEST INFRASTRUCTURE ONLY (parity checker + reported CPU baseline); see
 * rt_oracle.h.  Build with -ffp-contract=off: the reference is compiled
 * without fused multiply-add and every comparison below depends on it.
 *
 * Places where the reference's behaviour is undefined and this file fixes a
 * definition (mirrored by the GPU library):
 *   D1  texture paths: '\' -> '/', resolved against the .obj directory;
 *       absolute Windows paths fail to load (mesh.h:121-127).
 *   D2  mesh.cpp:499 can index one row past the texture; textures carry two
 *       extra rows of zero bytes.
 *   D3  (unsigned char)NaN is 0 (x86 cvttss2si), imageWriter.h:452.
 *   D5  Material fields never assigned by the .mtl (mesh.cpp:368 reuses one
 *       uninitialised Material) read as 0.
 *   D6  a face before any 'usemtl', or naming an unknown material, gets the
 *       built-in material 0 (mesh.cpp:314,335 dereference end()).
 */
#include "rt_oracle.h"

#include <ctype.h>
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ */
/* data                                                               */
/* ------------------------------------------------------------------ */

typedef struct {
    float Kd[3], Ka[3], Ks[3];
    float Ns, Ni, Tr;
    int   illum;
    int   Kd_set, Ka_set, Ks_set, Ns_set, Ni_set, Tr_set, illum_set;
    char  name[256];
    char  textureName[256];
    int   containsTexture;
    int   tex; /* index into scene textures, -1 none */
} or_material;

typedef struct {
    int            w, h;
    unsigned char *px; /* (h+2)*w*4 bytes, last two rows zero (D2) */
    char           path[512];
} or_texture_t;

typedef struct {
    unsigned v[3], t[3], n[3];
} or_triangle;

typedef struct {
    float b[6]; /* xStart xEnd yStart yEnd zStart zEnd (KDtree.h:23) */
    int   left, right, parent, depth;
    int   tri_off, tri_cnt;
} or_node;

struct or_scene {
    int          nv;
    float       *vpos, *vnrm;
    int          nt;
    or_triangle *tris;
    unsigned    *triMat;
    int          ntc;
    float       *texcoords;
    int          nm;
    or_material *mats;
    int          ntex;
    or_texture_t *tex;
    int          nl;
    float       *lights;
    int          nnodes, cap_nodes;
    or_node     *nodes;
    int          n_node_tris;
    int         *node_tris;
    int          treeDepth;
    int        **bl_lists; /* build-time: per-node triangle lists */
    int         *bl_counts;
    or_settings  set;
};

void or_default_settings(or_settings *s)
{
    s->backgroundlighting = 0.34f; /* settings.h:7 */
    s->diffusePower = 0.5f;        /* settings.h:8 */
    s->ShadowFactor = 0.5f;        /* settings.h:10 */
    s->MSAA = 1;                   /* settings.h:15 */
    s->depthPower = 2.5f;          /* settings.h:16 */
    s->backgroundColor[0] = 0.2f;  /* settings.h:18 */
    s->backgroundColor[1] = 0.5f;
    s->backgroundColor[2] = 1.0f;
    s->focusDepth = 0.0f;
    s->shadowSamples = 1;          /* settings.h:13 */
    s->shadowRadius = 0.9f;        /* settings.h:12 */
}

/* ------------------------------------------------------------------ */
/* Vec3D.h arithmetic                                                  */
/* ------------------------------------------------------------------ */

typedef struct {
    float p[3];
} v3;

static inline v3 v3m(float x, float y, float z)
{
    v3 r = {{x, y, z}};
    return r;
}
static inline v3 v3ld(const float *p) { return v3m(p[0], p[1], p[2]); }
static inline v3 vadd(v3 a, v3 b) { return v3m(a.p[0] + b.p[0], a.p[1] + b.p[1], a.p[2] + b.p[2]); }
static inline v3 vsub(v3 a, v3 b) { return v3m(a.p[0] - b.p[0], a.p[1] - b.p[1], a.p[2] - b.p[2]); }
/* Vec3D.h:12-18 */
static inline v3 vscale(v3 a, float f) { return v3m(a.p[0] * f, a.p[1] * f, a.p[2] * f); }
/* Vec3D.h:192-194 */
static inline float vdot(v3 a, v3 b) { return a.p[0] * b.p[0] + a.p[1] * b.p[1] + a.p[2] * b.p[2]; }
/* Vec3D.h:185-191 */
static inline v3 vcross(v3 a, v3 b)
{
    return v3m(a.p[1] * b.p[2] - a.p[2] * b.p[1], a.p[2] * b.p[0] - a.p[0] * b.p[2],
               a.p[0] * b.p[1] - a.p[1] * b.p[0]);
}
/* Vec3D.h:142-151; sqrt is the double overload there, rounded back to float */
static inline float vnormalize(v3 *a)
{
    float length = (float)sqrt((double)vdot(*a, *a));
    if (length == 0.0f) return 0;
    float rez = 1.0f / length;
    a->p[0] *= rez;
    a->p[1] *= rez;
    a->p[2] *= rez;
    return length;
}

/* ------------------------------------------------------------------ */
/* loader: mesh.cpp:96-480                                             */
/* ------------------------------------------------------------------ */

#define LINE_LEN 256 /* mesh.cpp:26 */

static void mat_cleanup(or_material *m) /* mesh.h:62-72 */
{
    m->Kd_set = m->Ka_set = m->Ks_set = m->Ns_set = m->Ni_set = m->Tr_set = m->illum_set = 0;
    strcpy(m->name, "empty");
}
static int mat_is_valid(const or_material *m) /* mesh.h:74-75 */
{
    return m->Kd_set || m->Ka_set || m->Ks_set || m->Tr_set;
}

static int find_mat(const or_scene *s, const char *key, int first)
{
    /* std::map<string,unsigned>: names are unique, index of the first push */
    for (int i = first; i < s->nm; ++i)
        if (strcmp(s->mats[i].name, key) == 0) return i;
    return -1;
}

static void push_mat(or_scene *s, const or_material *m)
{
    s->mats = (or_material *)realloc(s->mats, sizeof(or_material) * (size_t)(s->nm + 1));
    s->mats[s->nm++] = *m;
}

/* mesh.cpp:354-480 */
static void load_mtl(or_scene *s, const char *filename)
{
    FILE *in = fopen(filename, "r");
    if (!in) return;
    char        line[LINE_LEN];
    char        key[256] = "";
    or_material mat;
    memset(&mat, 0, sizeof mat); /* D5 */
    mat.tex = -1;
    mat_cleanup(&mat);
    int   indef = 0;
    float f1, f2, f3;
    memset(line, 0, LINE_LEN);
    while (!feof(in)) {
        if (!fgets(line, LINE_LEN, in)) { /* leaves the zeroed line */
        }
        if (line[0] == '#') {
            memset(line, 0, LINE_LEN);
            continue;
        } else if (isspace((unsigned char)line[0]) || line[0] == '\0') {
            if (indef && key[0] && mat_is_valid(&mat)) {
                if (find_mat(s, key, 1) < 0) {
                    strcpy(mat.name, key);
                    push_mat(s, &mat);
                }
                mat_cleanup(&mat);
            }
            if (line[0] == '\0') break;
        } else if (strncmp(line, "newmtl ", 7) == 0) {
            char *p0 = line + 6;
            while (isspace((unsigned char)*++p0)) {
            }
            char *p1 = p0;
            while (!isspace((unsigned char)*p1)) ++p1;
            *p1 = '\0';
            strncpy(key, p0, sizeof key - 1);
            indef = 1;
        } else if (strncmp(line, "Kd ", 3) == 0) {
            sscanf(line, "Kd %f %f %f", &f1, &f2, &f3);
            mat.Kd[0] = f1, mat.Kd[1] = f2, mat.Kd[2] = f3, mat.Kd_set = 1;
        } else if (strncmp(line, "Ka ", 3) == 0) {
            sscanf(line, "Ka %f %f %f", &f1, &f2, &f3);
            mat.Ka[0] = f1, mat.Ka[1] = f2, mat.Ka[2] = f3, mat.Ka_set = 1;
        } else if (strncmp(line, "Ks ", 3) == 0) {
            sscanf(line, "Ks %f %f %f", &f1, &f2, &f3);
            mat.Ks[0] = f1, mat.Ks[1] = f2, mat.Ks[2] = f3, mat.Ks_set = 1;
        } else if (strncmp(line, "Ns ", 3) == 0) {
            sscanf(line, "Ns %f", &f1);
            mat.Ns = f1, mat.Ns_set = 1;
        } else if (strncmp(line, "Ni ", 3) == 0) {
            sscanf(line, "Ni %f", &f1);
            mat.Ni = f1, mat.Ni_set = 1;
        } else if (strncmp(line, "illum ", 6) == 0) {
            int illum = -1;
            sscanf(line, "illum %i", &illum);
            mat.illum = illum, mat.illum_set = 1;
        } else if (strncmp(line, "map_Kd ", 7) == 0) {
            char *t = line + 7;
            size_t n = strlen(t);
            if (n && t[n - 1] == '\n') t[n - 1] = '\0';
            strncpy(mat.textureName, t, sizeof mat.textureName - 1);
        } else if (strncmp(line, "Tr ", 3) == 0) {
            sscanf(line, "Tr %f", &f1);
            mat.Tr = f1, mat.Tr_set = 1;
        } else if (strncmp(line, "d ", 2) == 0) {
            sscanf(line, "d %f", &f1);
            mat.Tr = f1, mat.Tr_set = 1;
        }
        if (feof(in) && indef && mat_is_valid(&mat) && key[0]) {
            if (find_mat(s, key, 1) < 0) {
                strcpy(mat.name, key);
                push_mat(s, &mat);
            }
        }
        memset(line, 0, LINE_LEN);
    }
    fclose(in);
}

static int load_rtex(const char *path, or_texture_t *t)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    unsigned char hdr[16];
    if (fread(hdr, 1, 16, f) != 16 || memcmp(hdr, "RTEX", 4) != 0) {
        fclose(f);
        return 0;
    }
    uint32_t w, h, zl;
    memcpy(&w, hdr + 4, 4), memcpy(&h, hdr + 8, 4), memcpy(&zl, hdr + 12, 4);
    unsigned char *z = (unsigned char *)malloc(zl);
    if (fread(z, 1, zl, f) != zl) {
        free(z);
        fclose(f);
        return 0;
    }
    fclose(f);
    uLongf rawlen = (uLongf)w * h * 4;
    t->px = (unsigned char *)calloc((size_t)(h + 2) * w * 4, 1);
    int rc = uncompress(t->px, &rawlen, z, zl);
    free(z);
    if (rc != Z_OK || rawlen != (uLongf)w * h * 4) {
        free(t->px);
        return 0;
    }
    t->w = (int)w, t->h = (int)h;
    return 1;
}

/* mesh.h:117-135 with D1/D2 */
static void load_texture(or_scene *s, or_material *m, const char *dir)
{
    m->tex = -1;
    if (m->textureName[0] == '\0') return;
    char rel[512];
    strncpy(rel, m->textureName, sizeof rel - 1);
    rel[sizeof rel - 1] = 0;
    for (char *p = rel; *p; ++p)
        if (*p == '\\') *p = '/';
    size_t n = strlen(rel);
    while (n && (rel[n - 1] == '\r' || rel[n - 1] == '\n')) rel[--n] = 0;
    if (rel[0] == '/' || (isalpha((unsigned char)rel[0]) && rel[1] == ':')) return; /* absolute */
    char path[1024];
    snprintf(path, sizeof path, "%s%s.rtex", dir, rel);
    for (int i = 0; i < s->ntex; ++i)
        if (strcmp(s->tex[i].path, path) == 0) {
            m->tex = i;
            m->containsTexture = 1;
            return;
        }
    or_texture_t t;
    memset(&t, 0, sizeof t);
    if (!load_rtex(path, &t)) return;
    strncpy(t.path, path, sizeof t.path - 1);
    s->tex = (or_texture_t *)realloc(s->tex, sizeof(or_texture_t) * (size_t)(s->ntex + 1));
    s->tex[s->ntex] = t;
    m->tex = s->ntex++;
    m->containsTexture = 1;
}

static or_scene *scene_new(void)
{
    or_scene *s = (or_scene *)calloc(1, sizeof *s);
    or_default_settings(&s->set);
    /* raytracing.cpp:38: one default light */
    s->nl = 1;
    s->lights = (float *)malloc(3 * sizeof(float));
    s->lights[0] = 0, s->lights[1] = 10, s->lights[2] = 0;
    return s;
}

/* mesh.cpp:96-351 */
or_scene *or_scene_load(const char *obj_path, int load_textures)
{
    FILE *in = fopen(obj_path, "r");
    if (!in) return NULL;
    or_scene *s = scene_new();

    /* mesh.cpp:110-119 built-in material 0 */
    or_material def;
    memset(&def, 0, sizeof def);
    def.tex = -1;
    mat_cleanup(&def);
    def.Kd[0] = def.Kd[1] = def.Kd[2] = 0.5f, def.Kd_set = 1;
    def.Ka_set = 1;
    def.Ks[0] = def.Ks[1] = def.Ks[2] = 0.5f, def.Ks_set = 1;
    def.Ns = 96.7f, def.Ns_set = 1;
    def.illum = 2, def.illum_set = 1;
    strcpy(def.name, "StandardMaterialInitFromTriMesh");
    push_mat(s, &def);

    /* directory of the obj, mesh.cpp:126-147 */
    char dir[512] = "";
    {
        const char *slash = strrchr(obj_path, '/');
        if (slash) {
            size_t n = (size_t)(slash - obj_path) + 1;
            memcpy(dir, obj_path, n);
            dir[n] = 0;
        }
    }

    char  line[LINE_LEN];
    char  matname[256] = "";
    int   cap_v = 0, cap_t = 0, cap_tc = 0;
    int   vh[64], th[64], nh[64];
    float x, y, z;
    memset(line, 0, LINE_LEN);
    while (!feof(in) && fgets(line, LINE_LEN, in)) {
        if (line[0] == '#' || isspace((unsigned char)line[0]) || line[0] == '\0') {
            memset(line, 0, LINE_LEN);
            continue;
        } else if (strncmp(line, "mtllib ", 7) == 0) {
            char *p0 = line + 6;
            while (isspace((unsigned char)*++p0)) {
            }
            char file[1024];
            size_t i = 0;
            while (p0[i] && !((unsigned char)p0[i] < 32 || (unsigned char)p0[i] == 255)) ++i;
            snprintf(file, sizeof file, "%s%.*s", dir, (int)i, p0);
            load_mtl(s, file);
        } else if (strncmp(line, "usemtl ", 7) == 0) {
            char *p0 = line + 6;
            while (isspace((unsigned char)*++p0)) {
            }
            char *p1 = p0;
            while (!isspace((unsigned char)*p1)) ++p1;
            *p1 = '\0';
            strncpy(matname, p0, sizeof matname - 1);
            if (find_mat(s, matname, 1) < 0) matname[0] = 0;
        } else if (strncmp(line, "v ", 2) == 0) {
            x = y = z = 0;
            sscanf(line, "v %f %f %f", &x, &y, &z);
            if (s->nv == cap_v) {
                cap_v = cap_v ? cap_v * 2 : 1024;
                s->vpos = (float *)realloc(s->vpos, sizeof(float) * 3 * (size_t)cap_v);
            }
            s->vpos[3 * s->nv] = x, s->vpos[3 * s->nv + 1] = y, s->vpos[3 * s->nv + 2] = z;
            s->nv++;
        } else if (strncmp(line, "vt ", 3) == 0) {
            float tc[3] = {0, 0, 0};
            sscanf(line, "vt %f %f", &tc[0], &tc[1]);
            if (s->ntc == cap_tc) {
                cap_tc = cap_tc ? cap_tc * 2 : 1024;
                s->texcoords = (float *)realloc(s->texcoords, sizeof(float) * 3 * (size_t)cap_tc);
            }
            memcpy(s->texcoords + 3 * s->ntc, tc, sizeof tc);
            s->ntc++;
        } else if (strncmp(line, "vn ", 3) == 0) {
            /* stored by the reference, never used for shading (mesh.cpp:213-219) */
        } else if (strncmp(line, "f ", 2) == 0) {
            /* mesh.cpp:221-341: '/'-separated components, missing ones skipped */
            int   nvh = 0, nth = 0, nnh = 0;
            char *p = line + 2;
            while (*p) {
                while (*p == ' ') ++p;
                if (*p == '\0' || *p == '\n' || *p == '\r') break;
                int component = 0;
                for (;;) {
                    char *q = p;
                    while (*q != '/' && *q != '\r' && *q != '\n' && *q != ' ' && *q != '\0') ++q;
                    char sep = *q;
                    if (q != p) {
                        char save = *q;
                        *q = '\0';
                        int val = atoi(p) - 1;
                        *q = save;
                        if (component == 0 && nvh < 64) vh[nvh++] = val;
                        else if (component == 1 && nth < 64) th[nth++] = val;
                        else if (component == 2 && nnh < 64) nh[nnh++] = val;
                    }
                    ++component;
                    p = q;
                    if (sep == '/') {
                        ++p;
                        continue;
                    }
                    break;
                }
                if (*p == '\0' || *p == '\n' || *p == '\r') break;
            }
            if (nvh != nth) {
                for (int i = nth; i < nvh; ++i) th[i] = 0;
                nth = nvh;
            }
            if (nvh != nnh) {
                for (int i = nnh; i < nvh; ++i) nh[i] = 0;
                nnh = nvh;
            }
            int m = matname[0] ? find_mat(s, matname, 1) : -1;
            if (m < 0) m = 0; /* D6 */
            int ntri = nvh >= 3 ? nvh - 2 : 0; /* fan (0,i+1,i+2), mesh.cpp:303-336 */
            for (int i = 0; i < ntri; ++i) {
                int a = 0, b = i + 1, c = i + 2;
                if (s->nt == cap_t) {
                    cap_t = cap_t ? cap_t * 2 : 1024;
                    s->tris = (or_triangle *)realloc(s->tris, sizeof(or_triangle) * (size_t)cap_t);
                    s->triMat = (unsigned *)realloc(s->triMat, sizeof(unsigned) * (size_t)cap_t);
                }
                or_triangle *t = &s->tris[s->nt];
                t->v[0] = (unsigned)vh[a], t->v[1] = (unsigned)vh[b], t->v[2] = (unsigned)vh[c];
                t->t[0] = (unsigned)th[a], t->t[1] = (unsigned)th[b], t->t[2] = (unsigned)th[c];
                t->n[0] = (unsigned)nh[a], t->n[1] = (unsigned)nh[b], t->n[2] = (unsigned)nh[c];
                s->triMat[s->nt] = (unsigned)m;
                s->nt++;
            }
        }
        memset(line, 0, LINE_LEN);
    }
    fclose(in);

    /* mesh.cpp:346-348: every material tries to load its texture */
    if (load_textures)
        for (int i = 0; i < s->nm; ++i) load_texture(s, &s->mats[i], dir);
    s->vnrm = (float *)calloc((size_t)s->nv * 3 + 3, sizeof(float));
    return s;
}

or_scene *or_scene_from_arrays(int nv, const float *vpos, int nt, const int32_t *tri_v,
                               const int32_t *tri_t, const int32_t *tri_mat, int ntc,
                               const float *texcoords, int nm, const float *materials)
{
    or_scene *s = scene_new();
    s->nv = nv;
    s->vpos = (float *)malloc(sizeof(float) * 3 * (size_t)(nv + 1));
    memcpy(s->vpos, vpos, sizeof(float) * 3 * (size_t)nv);
    s->vnrm = (float *)calloc((size_t)nv * 3 + 3, sizeof(float));
    s->nt = nt;
    s->tris = (or_triangle *)calloc((size_t)nt + 1, sizeof(or_triangle));
    s->triMat = (unsigned *)calloc((size_t)nt + 1, sizeof(unsigned));
    for (int i = 0; i < nt; ++i) {
        for (int j = 0; j < 3; ++j) {
            s->tris[i].v[j] = (unsigned)tri_v[3 * i + j];
            s->tris[i].t[j] = tri_t ? (unsigned)tri_t[3 * i + j] : 0;
        }
        s->triMat[i] = (unsigned)tri_mat[i];
    }
    s->ntc = ntc;
    s->texcoords = (float *)calloc((size_t)ntc * 3 + 3, sizeof(float));
    if (ntc) memcpy(s->texcoords, texcoords, sizeof(float) * 3 * (size_t)ntc);
    s->nm = nm;
    s->mats = (or_material *)calloc((size_t)nm, sizeof(or_material));
    for (int i = 0; i < nm; ++i) {
        const float *r = materials + 16 * i;
        or_material *m = &s->mats[i];
        memcpy(m->Kd, r, 12), memcpy(m->Ks, r + 3, 12);
        m->Ns = r[6], m->Ni = r[7], m->illum = (int)r[8], m->Ni_set = r[9] != 0.0f;
        m->tex = -1;
        snprintf(m->name, sizeof m->name, "m%d", i);
    }
    return s;
}

void or_scene_free(or_scene *s)
{
    if (!s) return;
    for (int i = 0; i < s->ntex; ++i) free(s->tex[i].px);
    free(s->tex), free(s->vpos), free(s->vnrm), free(s->tris), free(s->triMat);
    free(s->texcoords), free(s->mats), free(s->lights), free(s->nodes), free(s->node_tris);
    free(s);
}

void or_set_lights(or_scene *s, int n, const float *xyz)
{
    s->lights = (float *)realloc(s->lights, sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    memcpy(s->lights, xyz, sizeof(float) * 3 * (size_t)n);
    s->nl = n;
}
or_settings *or_scene_settings(or_scene *s) { return &s->set; }

/* mesh.cpp:32-52 */
void or_compute_vertex_normals(or_scene *s)
{
    memset(s->vnrm, 0, sizeof(float) * 3 * (size_t)s->nv);
    for (int i = 0; i < s->nt; ++i) {
        const or_triangle *t = &s->tris[i];
        v3 p0 = v3ld(s->vpos + 3 * t->v[0]), p1 = v3ld(s->vpos + 3 * t->v[1]),
           p2 = v3ld(s->vpos + 3 * t->v[2]);
        v3 n = vcross(vsub(p1, p0), vsub(p2, p0));
        vnormalize(&n);
        for (int j = 0; j < 3; ++j) {
            float *d = s->vnrm + 3 * t->v[j];
            d[0] += n.p[0], d[1] += n.p[1], d[2] += n.p[2];
        }
    }
    for (int i = 0; i < s->nv; ++i) {
        v3 n = v3ld(s->vnrm + 3 * i);
        vnormalize(&n);
        memcpy(s->vnrm + 3 * i, n.p, 12);
    }
}

int or_num_vertices(const or_scene *s) { return s->nv; }
int or_num_triangles(const or_scene *s) { return s->nt; }
int or_num_texcoords(const or_scene *s) { return s->ntc; }
int or_num_materials(const or_scene *s) { return s->nm; }
int or_num_textures(const or_scene *s) { return s->ntex; }
const float *or_vertex_pos(const or_scene *s) { return s->vpos; }
const float *or_vertex_nrm(const or_scene *s) { return s->vnrm; }
const float *or_texcoords(const or_scene *s) { return s->texcoords; }
void or_get_triangles(const or_scene *s, int32_t *v, int32_t *t, int32_t *mat)
{
    for (int i = 0; i < s->nt; ++i) {
        for (int j = 0; j < 3; ++j) {
            v[3 * i + j] = (int32_t)s->tris[i].v[j];
            t[3 * i + j] = (int32_t)s->tris[i].t[j];
        }
        mat[i] = (int32_t)s->triMat[i];
    }
}
void or_get_materials(const or_scene *s, float *out)
{
    for (int i = 0; i < s->nm; ++i) {
        const or_material *m = &s->mats[i];
        float *r = out + 16 * i;
        memset(r, 0, 64);
        memcpy(r, m->Kd, 12), memcpy(r + 3, m->Ks, 12);
        r[6] = m->Ns, r[7] = m->Ni, r[8] = (float)m->illum, r[9] = m->Ni_set ? 1.f : 0.f;
        r[10] = m->containsTexture ? (float)m->tex : -1.f;
    }
}
const char *or_material_name(const or_scene *s, int i) { return s->mats[i].name; }
const unsigned char *or_texture(const or_scene *s, int i, int *w, int *h)
{
    *w = s->tex[i].w, *h = s->tex[i].h;
    return s->tex[i].px;
}

/* ------------------------------------------------------------------ */
/* KD build: KDtree.cpp:15-230                                         */
/* ------------------------------------------------------------------ */

static int new_node(or_scene *s, const float b[6], int parent, int depth)
{
    if (s->nnodes == s->cap_nodes) {
        s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 256;
        s->nodes = (or_node *)realloc(s->nodes, sizeof(or_node) * (size_t)s->cap_nodes);
        s->bl_lists = (int **)realloc(s->bl_lists, sizeof(int *) * (size_t)s->cap_nodes);
        s->bl_counts = (int *)realloc(s->bl_counts, sizeof(int) * (size_t)s->cap_nodes);
    }
    s->bl_lists[s->nnodes] = NULL;
    s->bl_counts[s->nnodes] = 0;
    or_node *n = &s->nodes[s->nnodes];
    memcpy(n->b, b, sizeof n->b);
    n->left = n->right = -1;
    n->parent = parent;
    n->depth = depth;
    n->tri_off = n->tri_cnt = 0;
    return s->nnodes++;
}

/* KDtree.cpp:171-230 */
static float calc_best_split(const or_scene *s, const float b[6], const int *tri, int ntri, int axis)
{
    const int   A = axis % 3;
    const float TREE_ACCURACY = 1000.0f; /* KDtree.cpp:12 */
    float       mid, result = 0;
    float       cost = FLT_MAX, tempCost;
    const float start = b[2 * A], end = b[2 * A + 1];
    for (int i = 1; i < TREE_ACCURACY; ++i) {
        mid = ((end - start) * i / TREE_ACCURACY) + start;
        float  leftSize = mid - start, rightSize = end - mid, totalSize = end - start;
        size_t L = 0, R = 0, B = 0;
        for (int k = 0; k < ntri; ++k) {
            const or_triangle *t = &s->tris[tri[k]];
            float c0 = s->vpos[3 * t->v[0] + A], c1 = s->vpos[3 * t->v[1] + A],
                  c2 = s->vpos[3 * t->v[2] + A];
            if (c0 < mid && c1 < mid && c2 < mid) ++L;
            else if (c0 > mid && c1 > mid && c2 > mid) ++R;
            else ++B;
        }
        /* axis == 0 divides by zero: inf/NaN never beats FLT_MAX (KDtree.cpp:219) */
        tempCost = B * totalSize * 20.0f / axis + L * leftSize + R * rightSize;
        if (tempCost < cost) {
            cost = tempCost;
            result = mid;
        }
    }
    return result;
}

/* KDtree.cpp:75-169; tri is owned (freed here) */
static void split_spaces(or_scene *s, int node, int *tri, int ntri, int axis)
{
    if (axis > s->treeDepth) ++s->treeDepth;
    const int A = axis % 3;
    float     b[6];
    memcpy(b, s->nodes[node].b, sizeof b);
    if (ntri < 11 || (double)(b[2 * A + 1] - b[2 * A]) < 0.0001) {
        s->bl_lists[node] = tri;
        s->bl_counts[node] = ntri;
        return;
    }
    float mid = calc_best_split(s, b, tri, ntri, axis);
    float lb[6], rb[6];
    memcpy(lb, b, sizeof b), memcpy(rb, b, sizeof b);
    lb[2 * A + 1] = mid;
    rb[2 * A] = mid;
    int *lt = (int *)malloc(sizeof(int) * (size_t)(ntri + 1)), nl = 0;
    int *rt = (int *)malloc(sizeof(int) * (size_t)(ntri + 1)), nr = 0;
    int *bt = (int *)malloc(sizeof(int) * (size_t)(ntri + 1)), nb = 0;
    for (int k = 0; k < ntri; ++k) {
        const or_triangle *t = &s->tris[tri[k]];
        float c0 = s->vpos[3 * t->v[0] + A], c1 = s->vpos[3 * t->v[1] + A],
              c2 = s->vpos[3 * t->v[2] + A];
        if (c0 < mid && c1 < mid && c2 < mid) lt[nl++] = tri[k];
        else if (c0 > mid && c1 > mid && c2 > mid) rt[nr++] = tri[k];
        else bt[nb++] = tri[k];
    }
    free(tri);
    int depth = s->nodes[node].depth;
    int left = new_node(s, lb, node, depth + 1);
    int right = new_node(s, rb, node, depth + 1);
    s->nodes[node].left = left;
    s->nodes[node].right = right;
    s->bl_lists[node] = bt;
    s->bl_counts[node] = nb;
    split_spaces(s, left, lt, nl, axis + 1);
    split_spaces(s, right, rt, nr, axis + 1);
}

/* KDtree.cpp:15-73 */
void or_build_kdtree(or_scene *s)
{
    free(s->nodes), free(s->node_tris);
    s->nodes = NULL, s->node_tris = NULL, s->nnodes = s->cap_nodes = 0, s->treeDepth = 0;
    s->bl_lists = NULL, s->bl_counts = NULL;
    if (s->nt == 0) return;
    const or_triangle *t0 = &s->tris[0];
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) mn[a] = mx[a] = s->vpos[3 * t0->v[0] + a];
    for (int i = 0; i < s->nt; ++i)
        for (int j = 0; j < 3; ++j)
            for (int a = 0; a < 3; ++a) {
                float c = s->vpos[3 * s->tris[i].v[j] + a];
                if (c < mn[a]) mn[a] = c;
                else if (c > mx[a]) mx[a] = c;
            }
    float b[6];
    for (int a = 0; a < 3; ++a) b[2 * a] = mn[a] - 5.1f, b[2 * a + 1] = mx[a] + 5.1f;
    int  *all = (int *)malloc(sizeof(int) * (size_t)(s->nt + 1));
    for (int i = 0; i < s->nt; ++i) all[i] = i;
    int root = new_node(s, b, -1, 0);
    split_spaces(s, root, all, s->nt, 0);
    int **lists = s->bl_lists;
    int  *counts = s->bl_counts;
    int total = 0;
    for (int i = 0; i < s->nnodes; ++i) total += counts[i];
    s->node_tris = (int *)malloc(sizeof(int) * (size_t)(total + 1));
    s->n_node_tris = total;
    int off = 0;
    for (int i = 0; i < s->nnodes; ++i) {
        s->nodes[i].tri_off = off;
        s->nodes[i].tri_cnt = counts[i];
        memcpy(s->node_tris + off, lists[i], sizeof(int) * (size_t)counts[i]);
        off += counts[i];
        free(lists[i]);
    }
    free(lists), free(counts);
    s->bl_lists = NULL, s->bl_counts = NULL;
}

int or_num_nodes(const or_scene *s) { return s->nnodes; }
int or_tree_depth(const or_scene *s) { return s->treeDepth; }
int or_num_node_tris(const or_scene *s) { return s->n_node_tris; }
const int32_t *or_node_tris(const or_scene *s) { return s->node_tris; }
void or_get_nodes(const or_scene *s, float *bounds, int32_t *left, int32_t *right, int32_t *parent,
                  int32_t *depth, int32_t *tri_off, int32_t *tri_cnt)
{
    for (int i = 0; i < s->nnodes; ++i) {
        const or_node *n = &s->nodes[i];
        memcpy(bounds + 6 * i, n->b, 24);
        left[i] = n->left, right[i] = n->right, parent[i] = n->parent, depth[i] = n->depth;
        tri_off[i] = n->tri_off, tri_cnt[i] = n->tri_cnt;
    }
}

/* ------------------------------------------------------------------ */
/* KD walk: KDtree.cpp:235-605                                         */
/* ------------------------------------------------------------------ */

/* KDtree.cpp:235-266 */
static int find_child_node(const or_scene *s, int node, int axis, const float pos[3])
{
    while (s->nodes[node].left >= 0) {
        int A = axis % 3;
        const or_node *l = &s->nodes[s->nodes[node].left];
        if (l->b[2 * A + 1] > pos[A]) node = s->nodes[node].left;
        else node = s->nodes[node].right;
        ++axis;
    }
    return node;
}

/* KDtree.cpp:269-590.  Returns 0 (the root) when the walk leaves the box. */
static int find_next_node(const or_scene *s, int cur, const float pos[3], const float dst[3])
{
    const or_node *c = &s->nodes[cur];
    const or_node *root = &s->nodes[0];
    float dir[3] = {dst[0] - pos[0], dst[1] - pos[1], dst[2] - pos[2]};
    for (int A = 0; A < 3; ++A) {
        if (dir[A] == 0) continue;
        const int neg = dir[A] < 0;
        const float plane = neg ? c->b[2 * A] : c->b[2 * A + 1];
        const float f = (plane - pos[A]) / dir[A];
        float temp[3] = {0, 0, 0};
        const int B1 = (A + 1) % 3, B2 = (A + 2) % 3;
        temp[B1] = f * dir[B1] + pos[B1];
        temp[B2] = f * dir[B2] + pos[B2];
        if (A != 2) { /* the z faces are not range-checked (KDtree.cpp:491-587) */
            if (!(temp[B1] > c->b[2 * B1] && temp[B1] < c->b[2 * B1 + 1] &&
                  temp[B2] > c->b[2 * B2] && temp[B2] < c->b[2 * B2 + 1]))
                continue;
        }
        if (neg ? (c->b[2 * A] == root->b[2 * A]) : (c->b[2 * A + 1] == root->b[2 * A + 1]))
            return 0;
        int node = 0;
        for (int axis = 0;; ++axis) {
            const int ax = axis % 3;
            const or_node *n = &s->nodes[node];
            const float split = s->nodes[n->left].b[2 * ax + 1];
            int go_left;
            if (ax == A) {
                if (neg) go_left = (split >= c->b[2 * A]);
                else go_left = !(split <= c->b[2 * A + 1]);
            } else {
                const float t = temp[ax];
                if (t < split) go_left = 1;
                else if (t > split || dir[ax] > 0) go_left = 0;
                else go_left = 1;
            }
            node = go_left ? n->left : n->right;
            if (s->nodes[node].left < 0) return node;
        }
    }
    return 0;
}

/* KDtree.cpp:592-605: bounds equality against every node of the path */
static int path_contains(const or_scene *s, int leaf, const float hb[6])
{
    for (int n = leaf; n >= 0; n = s->nodes[n].parent) {
        const float *b = s->nodes[n].b;
        if (b[0] == hb[0] && b[2] == hb[2] && b[4] == hb[4] && b[1] == hb[1] && b[3] == hb[3] &&
            b[5] == hb[5])
            return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* ray/triangle: raytracing.cpp:354-386, Matrix33.h:124-139            */
/* ------------------------------------------------------------------ */

static inline float det33(v3 p0, v3 p1, v3 p2)
{
    return p0.p[0] * p1.p[1] * p2.p[2] + p1.p[0] * p2.p[1] * p0.p[2] + p2.p[0] * p0.p[1] * p1.p[2] -
           p0.p[2] * p1.p[1] * p2.p[0] - p1.p[2] * p2.p[1] * p0.p[0] - p2.p[2] * p0.p[1] * p1.p[0];
}

typedef struct {
    int   bHit;
    float a, b, t;
    v3    hitPoint;
} hitpair;

static inline hitpair tri_test(const or_scene *s, int tri, v3 origin, v3 dest, float minT)
{
    const or_triangle *T = &s->tris[tri];
    v3 dir = vsub(dest, origin);
    v3 v0 = v3ld(s->vpos + 3 * T->v[0]), v1 = v3ld(s->vpos + 3 * T->v[1]),
       v2 = v3ld(s->vpos + 3 * T->v[2]);
    v3 vec1 = vsub(v0, v2), vec2 = vsub(v1, v2);
    v3 c = vsub(origin, v2);
    float det = det33(vec1, vec2, dir);
    float r0, r1, r2;
    if (det == 0) r0 = r1 = r2 = 0.0f; /* Matrix33.h:132-134 */
    else {
        r0 = det33(c, vec2, dir) / det;
        r1 = det33(vec1, c, dir) / det;
        r2 = det33(vec1, vec2, c) / det;
    }
    hitpair h;
    h.a = r0, h.b = r1, h.t = -r2;
    h.bHit = !(h.a < 0 || h.b < 0 || h.a + h.b > 1 || h.t < 0 || h.t > minT);
    h.hitPoint = vadd(origin, vscale(dir, h.t));
    return h;
}

/* D7: the reference's leaf walk has no bound on its steps (raytracing.cpp:303-349); on a few rays its
 * neighbour search (KDtree.cpp:269-590) steps back and forth between leaves and never reaches the root
 * box, i.e. the reference would not return.  This restatement stops such a walk after WALK_GUARD steps
 * and keeps the closest hit found so far; the rays are logged (or_guard_rays) so that a fixture can
 * name them.  lego.obj at 3840x2160 MSAA 4 has two. */
#define WALK_GUARD 100000
#define GUARD_LOG 64
static float g_guard_log[GUARD_LOG][6];
static int   g_guard_n;

int or_guard_rays(float *out6, int max)
{
    int n = g_guard_n < GUARD_LOG ? g_guard_n : GUARD_LOG;
    if (n > max) n = max;
    if (out6 && n > 0) memcpy(out6, g_guard_log, sizeof(float) * 6 * (size_t)n);
    return g_guard_n;
}

void or_guard_reset(void) { g_guard_n = 0; }

/* raytracing.cpp:281-351 */
void or_check_hit(const or_scene *s, const float o[3], const float d[3], or_hit *out, or_counters *c)
{
    v3 origin = v3ld(o), dest = v3ld(d);
    float minT = FLT_MAX;
    out->bHit = 0, out->tri = -1, out->a = out->b = 0;
    out->hitPoint[0] = out->hitPoint[1] = out->hitPoint[2] = 0;
    if (s->nnodes == 0) {
        out->minT = minT;
        return;
    }
    int   cur = find_child_node(s, 0, 0, o);
    float hitBounds[6];
    memcpy(hitBounds, s->nodes[0].b, sizeof hitBounds); /* hitNode = treeRoot */
    int gotHit = 0;
    int oldPath[128], curPath[128], nOld = 0;
    uint64_t tests = 0, steps = 0;
    for (int guard = 0;; ++guard) {
        /* parentList of the current leaf: root ... leaf */
        int nCur = s->nodes[cur].depth + 1;
        for (int n = cur, i = nCur - 1; n >= 0; n = s->nodes[n].parent, --i) curPath[i] = n;
        int n = 0;
        while (n < nCur && n < nOld && curPath[n] == oldPath[n]) ++n;
        for (; n < nCur; ++n) {
            const or_node *nd = &s->nodes[curPath[n]];
            for (int i = 0; i < nd->tri_cnt; ++i) {
                int     tri = s->node_tris[nd->tri_off + i];
                hitpair h = tri_test(s, tri, origin, dest, minT);
                ++tests;
                if (!h.bHit) continue;
                memcpy(hitBounds, nd->b, sizeof hitBounds);
                gotHit = 1;
                out->bHit = 1, out->tri = tri, out->a = h.a, out->b = h.b;
                memcpy(out->hitPoint, h.hitPoint.p, 12);
                minT = h.t;
            }
        }
        if (gotHit && !path_contains(s, cur, hitBounds)) break;
        memcpy(oldPath, curPath, sizeof(int) * (size_t)nCur);
        nOld = nCur;
        cur = find_next_node(s, cur, o, d);
        ++steps;
        if (cur == 0) break; /* parentList.size() == 1 */
        if (guard > WALK_GUARD) {
            if (c) c->walk_guard++;
            const int at = __sync_fetch_and_add(&g_guard_n, 1);
            if (at < GUARD_LOG) memcpy(g_guard_log[at], o, 12), memcpy(g_guard_log[at] + 3, d, 12);
            break;
        }
    }
    out->minT = minT;
    if (c) c->tri_tests += tests, c->leaf_steps += steps;
}

/* ------------------------------------------------------------------ */
/* texture: mesh.cpp:485-553                                           */
/* ------------------------------------------------------------------ */

static v3 get_texture2(const or_texture_t *tx, float tex0, float tex1)
{
    if (tex0 > 0) tex0 = tex0 - (int)tex0;
    else tex0 = tex0 - (int)tex0 + 1;
    if (tex1 > 0) tex1 = tex1 - (int)tex1;
    else tex1 = tex1 - (int)tex1 + 1;
    int pos = tex0 * tx->w + (int)(tex1 * tx->h) * tx->w;
    const unsigned char *px = tx->px + (size_t)pos * 4;
    float red = px[0], green = px[1], blue = px[2];
    return v3m(red / 255.0f, green / 255.0f, blue / 255.0f);
}

static v3 get_texture(const or_scene *s, const or_material *m, const or_triangle *t, v3 hp)
{
    const float *tex0 = s->texcoords + 3 * t->t[0], *tex1 = s->texcoords + 3 * t->t[1],
                *tex2 = s->texcoords + 3 * t->t[2];
    float height, width;
    if ((double)fabsf(tex0[0] - tex2[0]) > 0.0001) width = hp.p[0] / (tex0[0] - tex2[0]);
    else if ((double)fabsf(tex0[0] - tex1[0]) > 0.0001) width = hp.p[0] / (tex0[0] - tex1[0]);
    else width = 0;
    if ((double)fabsf(tex0[1] - tex2[1]) > 0.0001) height = hp.p[1] / (tex0[1] - tex2[1]);
    else if ((double)fabsf(tex0[1] - tex1[1]) > 0.0001) height = hp.p[1] / (tex0[1] - tex1[1]);
    else height = 0;
    return get_texture2(&s->tex[m->tex], width, height);
}

/* ------------------------------------------------------------------ */
/* shading recursion: raytracing.cpp:61-274                            */
/* ------------------------------------------------------------------ */

/* Soft shadows (raytracing.cpp:145-160) draw three rand() values per shadow sample.  rand()
 * shared by up to 16 threads has no reproducible order, so its stream is NOT restated; the
 * stand-in is a counter-based generator keyed by where the sample sits: id = 32 * (index of the
 * camera ray: y * W + x for a frame, the ray's index for a batch) + sample number (0 = the
 * pixel centre, 1.. = MSAA samples); path = the route from the camera ray to this hit, two bits
 * per level (1 mirror, 2 transmitted, 3 reflected child); the light, the sample's ordinal in the
 * x, y, z loops and the component.  It returns 0 .. 2^31 - 1, the range of rand() here. */
uint32_t or_soft_rand(uint64_t id, uint32_t path, uint32_t light, uint32_t sample, uint32_t comp)
{
    uint64_t z = id * 0x9E3779B97F4A7C15ull +
                 (((uint64_t)path << 24) | ((uint64_t)light << 16) | ((uint64_t)sample << 2) | comp);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 33);
}

static v3 perform(const or_scene *s, v3 origin, v3 dest, int k, float prev_Ni, float *minT,
                  int *firstTri, or_counters *c, uint64_t id, uint32_t path)
{
    if (k > 5) return v3m(0, 0, 0);
    v3 resCol;
    v3 dir = vsub(dest, origin);
    or_hit hit;
    or_check_hit(s, origin.p, dest.p, &hit, c);
    if (c) c->rays_closest++;
    *minT = hit.minT;
    if (firstTri) *firstTri = hit.tri;
    if (!hit.bHit) return v3ld(s->set.backgroundColor);
    if (c) c->shaded_hits++;

    const or_triangle *tri = &s->tris[hit.tri];
    const or_material *mat = &s->mats[s->triMat[hit.tri]];
    v3 hitPoint = v3ld(hit.hitPoint);
    v3 nA = v3ld(s->vnrm + 3 * tri->v[0]), nB = v3ld(s->vnrm + 3 * tri->v[1]),
       nC = v3ld(s->vnrm + 3 * tri->v[2]);
    float u = hit.a, v = hit.b, w = 1.f - u - v;
    v3 N = vadd(vadd(vscale(nA, u), vscale(nB, v)), vscale(nC, w));
    vnormalize(&N);

    if (!mat->containsTexture) resCol = vscale(v3ld(mat->Kd), s->set.backgroundlighting);
    else resCol = get_texture(s, mat, tri, hitPoint);

    float specularHighlight = mat->Ns;
    float shadowPart = s->set.ShadowFactor / s->nl;
    v3    shadowRGB = vscale(v3m(1.f, 1.f, 1.f), shadowPart);
    float shadowMax = 0.f;
    float smFree = 0.f; /* counters only: shadowMax if every query so far found the light free */
    v3    viewDir = vsub(origin, hitPoint);
    vnormalize(&viewDir);

    for (int li = 0; li < s->nl; ++li) {
        if (c) c->hit_lights++;
        v3 lpos = v3ld(s->lights + 3 * li);
        v3 lightDir = vsub(lpos, hitPoint);
        vnormalize(&lightDir);
        float angle = vdot(N, lightDir);
        if (angle > 0) resCol = vadd(resCol, vscale(vscale(v3ld(mat->Kd), angle), s->set.diffusePower));
        angle = vdot(lightDir, N);
        if (angle <= 0.2f) shadowMax = smFree = 1.f;
        else if ((angle > 0.2f) & (angle < 0.5f)) {
            angle -= 0.2f;
            shadowMax = smFree = (3.33f * (0.3f - angle));
        }
        float softShadow = 0.f;
        if (angle > 0.f) {
            v3 halfway = vadd(lightDir, viewDir);
            vnormalize(&halfway);
            angle = vdot(N, halfway);
            if (angle > 0.f) resCol = vadd(resCol, vscale(v3ld(mat->Ks), powf(angle, specularHighlight)));
            or_hit sh;
            v3 so = vadd(hitPoint, vscale(lightDir, 0.001f));
            const int shadowSamples = s->set.shadowSamples;
            if (shadowSamples <= 1) {
                uint64_t t0 = c ? c->tri_tests : 0, s0 = c ? c->leaf_steps : 0;
                or_check_hit(s, so.p, lpos.p, &sh, c);
                if (c) c->rays_shadow++;
                if (c && smFree == 1.f) {
                    c->moot_shadow_rays++;
                    c->moot_tri_tests += c->tri_tests - t0;
                    c->moot_leaf_steps += c->leaf_steps - s0;
                }
                if (sh.bHit) shadowMax = 1.f;
            }
            if (shadowSamples > 1) { /* raytracing.cpp:145-160 */
                const float shadowRadius = s->set.shadowRadius;
                float    delta = 2.f / (float)(shadowSamples - 1 + 0.000000000000000000001f);
                uint32_t si = 0;
                for (float x = -1.f; x <= 1.f; x += delta)
                    for (float y = -1.f; y <= 1.f; y += delta)
                        for (float z = -1.f; z <= 1.f; z += delta) {
                            const double RM = 2147483647.0; /* RAND_MAX */
                            float rx = delta / 2.f * (double)or_soft_rand(id, path, li, si, 0) / RM - delta / 4.f;
                            float ry = delta / 2.f * (double)or_soft_rand(id, path, li, si, 1) / RM - delta / 4.f;
                            float rz = delta / 2.f * (double)or_soft_rand(id, path, li, si, 2) / RM - delta / 4.f;
                            ++si;
                            v3 sd = vadd(lpos, v3m(x * shadowRadius + rx, y * shadowRadius + ry,
                                                   z * shadowRadius + rz));
                            or_check_hit(s, so.p, sd.p, &sh, c);
                            if (c) c->rays_shadow++;
                            if (sh.bHit) softShadow += 1.0f / shadowSamples / shadowSamples / shadowSamples;
                        }
            }
        }
        shadowMax = (shadowMax < softShadow) ? softShadow : shadowMax; /* std::max(shadowMax, softShadow) */
        smFree = (smFree < softShadow) ? softShadow : smFree;
        resCol = vsub(resCol, vscale(shadowRGB, shadowMax));
    }

    int newK = k + 1;
    if (newK >= 5) return resCol;

    float n2 = mat->Ni_set ? mat->Ni : 1.0f;
    float n1 = (prev_Ni != 0.0f) ? prev_Ni : 1.0f;

    if (mat->illum == 3) {
        v3 reflectdir = vsub(dir, vscale(N, 2 * vdot(N, dir)));
        v3 newOrigin = vadd(hitPoint, vscale(reflectdir, 0.001f));
        v3 newDest = vadd(hitPoint, reflectdir);
        float dummy;
        if (c) c->child_rays++;
        v3 newCol = perform(s, newOrigin, newDest, newK, n2, &dummy, NULL, c, id, path * 4 + 1);
        return vadd(vscale(resCol, (float)0.5), vscale(newCol, (float)0.5));
    }

    v3 vInc = dir;
    vnormalize(&vInc);
    if (mat->Ni > 1.0f) {
        float refractIndex = (n1 != n2) ? n2 : (1 / n2);
        float cosI = vdot(vInc, N);
        /* pow(float,int) promotes to double (raytracing.cpp:221) */
        float sin2T = (float)(((double)refractIndex * (double)refractIndex) *
                              (1 - ((double)cosI * (double)cosI)));
        v3 newColT = v3m(0, 0, 0), newColR;
        float dummy;
        if (sin2T <= 1.0f) {
            /* sqrt resolves to the double overload in that translation unit */
            float f = (float)((double)(refractIndex * cosI) - sqrt((double)(1 - sin2T)));
            v3 tRay = vadd(vscale(vInc, refractIndex), vscale(N, f));
            v3 newOriginT = vadd(hitPoint, vscale(tRay, 0.001f));
            v3 newDestT = vadd(hitPoint, tRay);
            if (c) c->child_rays++;
            newColT = perform(s, newOriginT, newDestT, newK, n2, &dummy, NULL, c, id, path * 4 + 2);
            newColT = vadd(vscale(resCol, (float)0.1), vscale(newColT, (float)0.9));
        }
        v3 refRay = vsub(vInc, vscale(N, 2 * vdot(vInc, N)));
        v3 newOriginR = vadd(hitPoint, vscale(refRay, 0.001f));
        v3 newDestR = vadd(hitPoint, refRay);
        if (c) c->child_rays++;
        newColR = perform(s, newOriginR, newDestR, newK, n2, &dummy, NULL, c, id, path * 4 + 3);
        newColR = vadd(vscale(resCol, (float)0.5), vscale(newColR, (float)0.5));
        float rf = sin2T;
        rf = (rf < 0 ? 0 : rf);
        rf = (rf > 1 ? 1 : rf);
        resCol = vadd(vscale(newColR, rf), vscale(newColT, (1 - rf)));
    }
    return resCol;
}

void or_perform_ray_tracing(const or_scene *s, const float o[3], const float d[3], int k,
                            float prev_Ni, float *minT, float rgb[3], or_counters *c)
{
    v3 r = perform(s, v3ld(o), v3ld(d), k, prev_Ni, minT, NULL, c, 0, 0);
    memcpy(rgb, r.p, 12);
}

/* ------------------------------------------------------------------ */
/* threading helper                                                    */
/* ------------------------------------------------------------------ */

typedef struct job job;
typedef void (*job_fn)(job *, int idx, or_counters *c);
struct job {
    job_fn       fn;
    int          n, chunk;
    volatile int next;
    void        *arg;
    or_counters *total;
    pthread_mutex_t mu;
};

static void counters_add(or_counters *a, const or_counters *b)
{
    uint64_t *x = (uint64_t *)a;
    const uint64_t *y = (const uint64_t *)b;
    for (size_t i = 0; i < sizeof(or_counters) / sizeof(uint64_t); ++i) x[i] += y[i];
}

static void *job_worker(void *p)
{
    job *j = (job *)p;
    or_counters local;
    memset(&local, 0, sizeof local);
    for (;;) {
        int start = __atomic_fetch_add(&j->next, j->chunk, __ATOMIC_RELAXED);
        if (start >= j->n) break;
        int end = start + j->chunk < j->n ? start + j->chunk : j->n;
        for (int i = start; i < end; ++i) j->fn(j, i, &local);
    }
    pthread_mutex_lock(&j->mu);
    if (j->total) counters_add(j->total, &local);
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

static void run_job(job_fn fn, int n, int chunk, void *arg, int threads, or_counters *total)
{
    job j;
    j.fn = fn, j.n = n, j.chunk = chunk > 0 ? chunk : 1, j.next = 0, j.arg = arg, j.total = total;
    pthread_mutex_init(&j.mu, NULL);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (threads == 1) {
        job_worker(&j);
    } else {
        pthread_t th[256];
        for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, job_worker, &j);
        for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
    }
    pthread_mutex_destroy(&j.mu);
}

typedef struct {
    const or_scene *s;
    const float    *o, *d;
    float          *rgb, *minT, *a, *b, *t, *hp;
    int32_t        *tri;
} rays_arg;

static void trace_one(job *j, int i, or_counters *c)
{
    rays_arg *A = (rays_arg *)j->arg;
    float     mt;
    int       ft = -1;
    v3 r = perform(A->s, v3ld(A->o + 3 * i), v3ld(A->d + 3 * i), 0, 1.0f, &mt, &ft, c, (uint64_t)i * 32, 0);
    memcpy(A->rgb + 3 * i, r.p, 12);
    A->minT[i] = mt;
    if (A->tri) A->tri[i] = ft;
}

void or_trace_rays(const or_scene *s, int n, const float *origins, const float *dests, float *rgb,
                   float *minT, int32_t *tri, int threads, or_counters *c)
{
    rays_arg A = {s, origins, dests, rgb, minT, NULL, NULL, NULL, NULL, tri};
    run_job(trace_one, n, 256, &A, threads, c);
}

static void hit_one(job *j, int i, or_counters *c)
{
    rays_arg *A = (rays_arg *)j->arg;
    or_hit    h;
    or_check_hit(A->s, A->o + 3 * i, A->d + 3 * i, &h, c);
    if (c) c->rays_closest++;
    A->tri[i] = h.tri;
    if (A->a) A->a[i] = h.a;
    if (A->b) A->b[i] = h.b;
    if (A->t) A->t[i] = h.minT;
    if (A->hp) memcpy(A->hp + 3 * i, h.hitPoint, 12);
}

void or_check_hits(const or_scene *s, int n, const float *origins, const float *dests, int32_t *tri,
                   float *a, float *b, float *t, float *hitPoint, int threads, or_counters *c)
{
    rays_arg A = {s, origins, dests, NULL, NULL, a, b, t, hitPoint, tri};
    run_job(hit_one, n, 256, &A, threads, c);
}

void or_brute_hits(const or_scene *s, int n, const float *origins, const float *dests, int32_t *tri,
                   float *t)
{
    for (int i = 0; i < n; ++i) {
        float minT = FLT_MAX;
        int   best = -1;
        v3    o = v3ld(origins + 3 * i), d = v3ld(dests + 3 * i);
        for (int k = 0; k < s->nt; ++k) {
            hitpair h = tri_test(s, k, o, d, minT);
            if (h.bHit) best = k, minT = h.t;
        }
        tri[i] = best;
        t[i] = minT;
    }
}

/* ------------------------------------------------------------------ */
/* frame driver: main.cpp:43-56, 383-388, 435-552                      */
/* ------------------------------------------------------------------ */

static int invert4(const double m[16], double inv[16])
{
    double a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
           m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
           m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
           m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
            m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
           m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
           m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
           m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
            m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
           m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
           m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
            m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
            m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
           m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
           m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
            m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
            m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
    if (det == 0) return 0;
    det = 1.0 / det;
    for (int i = 0; i < 16; ++i) inv[i] = a[i] * det;
    return 1;
}

/* gluUnProject as called from main.cpp:544,548 (row-vector convention on the
 * column-major arrays, like the GLU sample implementation) */
static void unproject(double winx, double winy, double winz, const double model[16],
                      const double proj[16], const int vp[4], double out3[3])
{
    double fin[16], inv[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            fin[i * 4 + j] = model[i * 4 + 0] * proj[0 * 4 + j] + model[i * 4 + 1] * proj[1 * 4 + j] +
                             model[i * 4 + 2] * proj[2 * 4 + j] + model[i * 4 + 3] * proj[3 * 4 + j];
    if (!invert4(fin, inv)) {
        out3[0] = out3[1] = out3[2] = 0;
        return;
    }
    double in[4] = {winx, winy, winz, 1.0}, out[4];
    in[0] = (in[0] - vp[0]) / vp[2];
    in[1] = (in[1] - vp[1]) / vp[3];
    in[0] = in[0] * 2 - 1, in[1] = in[1] * 2 - 1, in[2] = in[2] * 2 - 1;
    for (int i = 0; i < 4; ++i)
        out[i] = in[0] * inv[0 * 4 + i] + in[1] * inv[1 * 4 + i] + in[2] * inv[2 * 4 + i] +
                 in[3] * inv[3 * 4 + i];
    if (out[3] == 0.0) {
        out3[0] = out3[1] = out3[2] = 0;
        return;
    }
    out3[0] = out[0] / out[3], out3[1] = out[1] / out[3], out3[2] = out[2] / out[3];
}

/* main.cpp:52-55 + 532-552 with the projection of main.cpp:209 */
void or_camera_corners(const double modelview[16], int W, int H, float corners[24])
{
    double proj[16];
    memset(proj, 0, sizeof proj);
    const double fovy = 50, zNear = 0.01, zFar = 10;
    const double aspect = (double)((float)W / H);
    const double radians = fovy / 2 * 3.14159265358979323846 / 180;
    const double deltaZ = zFar - zNear, sine = sin(radians);
    const double cotangent = cos(radians) / sine;
    proj[0] = cotangent / aspect;
    proj[5] = cotangent;
    proj[10] = -(zFar + zNear) / deltaZ;
    proj[11] = -1;
    proj[14] = -2 * zNear * zFar / deltaZ;
    proj[15] = 0;
    const int vp[4] = {0, 0, W, H};
    const int px[4][2] = {{0, 0}, {0, H - 1}, {W - 1, 0}, {W - 1, H - 1}}; /* 00 01 10 11 */
    for (int c = 0; c < 4; ++c) {
        int    y_new = vp[3] - px[c][1];
        double o[3], d[3];
        unproject(px[c][0], y_new, 0, modelview, proj, vp, o);
        unproject(px[c][0], y_new, 1, modelview, proj, vp, d);
        for (int a = 0; a < 3; ++a) {
            corners[(2 * c) * 3 + a] = (float)o[a];
            corners[(2 * c + 1) * 3 + a] = (float)d[a];
        }
    }
}

/* main.cpp:514-526 */
void or_pixel_ray(const float corners[24], int W, int H, double x, double y, float o[3], float d[3])
{
    double xscale = 1.0 - x / (double)(W - 1);
    double yscale = 1.0 - y / (double)(H - 1);
    float  xs = (float)xscale, xs1 = (float)(1 - xscale), ys = (float)yscale, ys1 = (float)(1 - yscale);
    v3 o00 = v3ld(corners), d00 = v3ld(corners + 3), o01 = v3ld(corners + 6), d01 = v3ld(corners + 9),
       o10 = v3ld(corners + 12), d10 = v3ld(corners + 15), o11 = v3ld(corners + 18),
       d11 = v3ld(corners + 21);
    v3 ro = vadd(vscale(vadd(vscale(o00, xs), vscale(o10, xs1)), ys),
                 vscale(vadd(vscale(o01, xs), vscale(o11, xs1)), ys1));
    v3 rd = vadd(vscale(vadd(vscale(d00, xs), vscale(d10, xs1)), ys),
                 vscale(vadd(vscale(d01, xs), vscale(d11, xs1)), ys1));
    memcpy(o, ro.p, 12), memcpy(d, rd.p, 12);
}

static v3 raytrace(const or_scene *s, const float corners[24], int W, int H, double x, double y,
                   float *depth, int *tri, or_counters *c, uint64_t id)
{
    float o[3], d[3];
    or_pixel_ray(corners, W, H, x, y, o, d);
    if (c) c->camera_rays++;
    return perform(s, v3ld(o), v3ld(d), 0, 1.0f, depth, tri, c, id, 0);
}

static void counters_add(or_counters *a, const or_counters *b);

/* main.cpp:383-388 */
static void depth_increase(float *depth, float holder, int *hitNumber)
{
    if (holder != FLT_MAX) {
        *depth += holder;
        ++*hitNumber;
    }
}

static const double MSAA4[4][2] = {{-1. / 8., -3. / 8.}, {3. / 8., -1. / 8.}, {-3. / 8., 1. / 8.},
                                   {1. / 8., 3. / 8.}}; /* main.cpp:452-458 */
static const double MSAA16[16][2] = {
    {-9. / 32., -15. / 32.}, {-1. / 32., -13. / 32.}, {7. / 32., -11. / 32.}, {15. / 32., -9. / 32.},
    {-11. / 32., -7. / 32.}, {-3. / 32., -5. / 32.},  {5. / 32., -3. / 32.},  {13. / 32., -1. / 32.},
    {-13. / 32., 1. / 32.},  {-5. / 32., 3. / 32.},   {3. / 32., 5. / 32.},   {11. / 32., 7. / 32.},
    {-15. / 32., 9. / 32.},  {-7. / 32., 11. / 32.},  {1. / 32., 13. / 32.},  {9. / 32., 15. / 32.}};
/* main.cpp:469-500 */

/* main.cpp:435-511 */
static v3 raytrace_msaa(const or_scene *s, const float corners[24], int W, int H, int x, int y,
                        float *depth, int *primaryTri, or_counters *c, or_counters *centre)
{
    int   hitNumber = 0;
    float holder;
    *depth = 0;
    or_counters cc;
    memset(&cc, 0, sizeof cc);
    const uint64_t id0 = ((uint64_t)y * (uint64_t)W + (uint64_t)x) * 32;
    v3 rgb = raytrace(s, corners, W, H, x, y, &holder, primaryTri, &cc, id0);
    if (c) counters_add(c, &cc);
    if (centre) counters_add(centre, &cc);
    depth_increase(depth, holder, &hitNumber);
    if (hitNumber == 0) *depth = FLT_MAX;
    if (hitNumber > 0 && (s->set.MSAA == 4 || s->set.MSAA == 16)) {
        const int n = s->set.MSAA;
        const double(*off)[2] = n == 4 ? MSAA4 : MSAA16;
        *depth = 0;
        hitNumber = 0;
        for (int i = 0; i < n; ++i) {
            v3 r = raytrace(s, corners, W, H, x + off[i][0], y + off[i][1], &holder, NULL, c, id0 + 1 + i);
            rgb = i == 0 ? r : vadd(rgb, r);
            depth_increase(depth, holder, &hitNumber);
        }
        if (n == 4) rgb = v3m(rgb.p[0] / 4, rgb.p[1] / 4, rgb.p[2] / 4); /* Vec3D /= T */
        else { /* rgb /= 16. : T is float, so the divisor is converted first */
            float dv = (float)16.;
            rgb = v3m(rgb.p[0] / dv, rgb.p[1] / dv, rgb.p[2] / dv);
        }
        if (hitNumber > 0) *depth /= hitNumber;
        else *depth = FLT_MAX;
    }
    return rgb;
}

static inline float clamp01(float v) /* imageWriter.h:26-44 */
{
    if (v > 1) v = 1.0;
    if (v < 0) v = 0.0;
    return v;
}

typedef struct {
    const or_scene *s;
    const float    *corners;
    int             W, H, y0, chunks;
    float          *rgb, *depth;
    int32_t        *tri;
    or_counters    *centre;
    pthread_mutex_t mu;
} rows_arg;

#define ROW_CHUNK 64

/* one work item = ROW_CHUNK pixels of one row */
static void render_row(job *j, int i, or_counters *c)
{
    rows_arg *A = (rows_arg *)j->arg;
    int       y = A->y0 + i / A->chunks;
    int       x0 = (i % A->chunks) * ROW_CHUNK, x1 = x0 + ROW_CHUNK < A->W ? x0 + ROW_CHUNK : A->W;
    or_counters cen;
    memset(&cen, 0, sizeof cen);
    for (int x = x0; x < x1; ++x) {
        float depth;
        int   tri = -1;
        v3 rgb = raytrace_msaa(A->s, A->corners, A->W, A->H, x, y, &depth, &tri, c,
                               A->centre ? &cen : NULL);
        size_t p = (size_t)y * A->W + x;
        A->rgb[3 * p] = clamp01(rgb.p[0]);
        A->rgb[3 * p + 1] = clamp01(rgb.p[1]);
        A->rgb[3 * p + 2] = clamp01(rgb.p[2]);
        A->depth[p] = depth;
        if (A->tri) A->tri[p] = tri;
    }
    if (A->centre) {
        pthread_mutex_lock(&A->mu);
        counters_add(A->centre, &cen);
        pthread_mutex_unlock(&A->mu);
    }
}

void or_render_rows2(const or_scene *s, const float corners[24], int W, int H, int y0, int y1,
                     float *rgb, float *depth, int32_t *primary_tri, int threads, or_counters *c,
                     or_counters *centre)
{
    rows_arg A = {s, corners, W, H, y0, (W + ROW_CHUNK - 1) / ROW_CHUNK, rgb, depth, primary_tri,
                  centre, PTHREAD_MUTEX_INITIALIZER};
    run_job(render_row, (y1 - y0) * A.chunks, 1, &A, threads, c);
}

void or_render_rows(const or_scene *s, const float corners[24], int W, int H, int y0, int y1,
                    float *rgb, float *depth, int32_t *primary_tri, int threads, or_counters *c)
{
    or_render_rows2(s, corners, W, H, y0, y1, rgb, depth, primary_tri, threads, c, NULL);
}

/* ------------------------------------------------------------------ */
/* image: imageWriter.h                                                */
/* ------------------------------------------------------------------ */

/* imageWriter.h:113-135 */
void or_normalize_depth(float *depth, int n, float focusDepth)
{
    float lowest = FLT_MAX, highest = 0;
    for (int i = 0; i < n; ++i) {
        lowest = lowest < depth[i] ? lowest : depth[i];
        if (depth[i] != FLT_MAX) highest = highest > depth[i] ? highest : depth[i];
    }
    float a = fabsf(lowest - focusDepth), b = fabsf(highest - focusDepth);
    float denom = a > b ? a : b;
    for (int i = 0; i < n; ++i) {
        if (depth[i] != FLT_MAX) depth[i] = fabsf(depth[i] - focusDepth) / denom;
        else depth[i] = 0;
    }
}

/* imageWriter.h:137-181 */
void or_gaussian_map(float intensity, float depthPower, float map[169])
{
    intensity = 1 - intensity;
    intensity *= intensity;
    float total = 0.;
    if (intensity >= 1) {
        for (int i = 0; i < 169; ++i) map[i] = 0.0f;
        map[6 * 13 + 6] = 1;
        return;
    }
    for (int i = 0; i < 13; i++)
        for (int j = 0; j < 13; j++) {
            /* integer division by 2, then times (intensity*depthPower)^2 */
            float ex = (float)((-(i - 6) * (i - 6) - (j - 6) * (j - 6)) / 2) *
                       powf(intensity * depthPower, 2);
            map[i * 13 + j] =
                (float)(1 / ((double)(intensity * depthPower * 2) * 3.14159265358979323846) *
                        (double)powf((float)2.71828182845904523536, ex));
            total += map[i * 13 + j];
        }
    for (int i = 0; i < 169; ++i) map[i] /= total;
}

typedef struct {
    const float *rgb, *depthN;
    int          W, H;
    float        depthPower;
    float       *out;
} blur_arg;

/* imageWriter.h:210-231 */
static void blur_row(job *j, int idx, or_counters *c)
{
    (void)c;
    blur_arg *A = (blur_arg *)j->arg;
    int       y = idx + 6, W = A->W, bw = W - 12;
    float     map[169];
    for (int i = 6; i < W - 6; i++) {
        float r = 0.0f, g = 0.0f, b = 0.0f;
        or_gaussian_map(A->depthN[y * W + i], A->depthPower, map);
        for (int jj = 0; jj < 13; jj++)
            for (int k = 0; k < 13; k++) {
                const float *p = A->rgb + 3 * ((size_t)(y + jj - 6) * W + i + k - 6);
                r += p[0] * map[jj * 13 + k];
                g += p[1] * map[jj * 13 + k];
                b += p[2] * map[jj * 13 + k];
            }
        float *o = A->out + 3 * ((size_t)(y - 6) * bw + i - 6);
        o[0] = r, o[1] = g, o[2] = b;
    }
}

void or_blur(const float *rgb, const float *depthN, int W, int H, float depthPower, float *out,
             int threads)
{
    blur_arg A = {rgb, depthN, W, H, depthPower, out};
    if (H - 12 > 0 && W - 12 > 0) run_job(blur_row, H - 12, 1, &A, threads, NULL);
}

/* (unsigned char)(v*255.0f); NaN -> 0 (D3) */
static inline unsigned char to_byte(float v)
{
    float x = v * 255.0f;
    if (x != x) return 0;
    if (x >= 2147483648.0f || x < -2147483648.0f) return 0; /* cvttss2si indefinite */
    return (unsigned char)(int)x;
}
void or_to_bytes(const float *v, int n, unsigned char *out)
{
    for (int i = 0; i < n; ++i) out[i] = to_byte(v[i]);
}

static void bmp_headers(unsigned char fh[14], unsigned char ih[40], unsigned sizeField, int w, int h)
{
    static const unsigned char f0[14] = {'B', 'M', 0, 0, 0, 0, 0, 0, 0, 0, 54, 0, 0, 0};
    static const unsigned char i0[40] = {40, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 24, 0};
    memcpy(fh, f0, 14), memcpy(ih, i0, 40);
    for (int i = 0; i < 4; ++i) {
        fh[2 + i] = (unsigned char)(sizeField >> (8 * i));
        ih[4 + i] = (unsigned char)((unsigned)w >> (8 * i));
        ih[8 + i] = (unsigned char)((unsigned)(-h) >> (8 * i));
    }
}

/* imageWriter.h:283-340 (whole-image form) */
int or_write_bmp(const char *path, const float *rgb, int W, int H)
{
    FILE *f = fopen(path, "wb");
    if (!f) return 0;
    unsigned char fh[14], ih[40];
    bmp_headers(fh, ih, (unsigned)(3 * W * H), W, H);
    fwrite(fh, 1, 14, f), fwrite(ih, 1, 40, f);
    int rowsize = ((24 * W + 31) / 32) * 4;
    unsigned char *buf = (unsigned char *)calloc((size_t)rowsize * H + 4, 1);
    for (int y = 0; y < H; ++y) {
        int i = 0;
        for (; i < W * 3; i += 3) {
            buf[y * rowsize + i] = to_byte(rgb[y * W * 3 + i + 2]);
            buf[y * rowsize + i + 1] = to_byte(rgb[y * W * 3 + i + 1]);
            buf[y * rowsize + i + 2] = to_byte(rgb[y * W * 3 + i]);
        }
        for (; i < rowsize; ++i) buf[y * rowsize + i] = 1;
    }
    size_t t = fwrite(buf, (size_t)W * H * 3, 1, f);
    free(buf);
    fclose(f);
    return t == 1;
}

/* imageWriter.h:253-281 */
int or_write_ppm(const char *path, const float *rgb, int W, int H)
{
    FILE *f = fopen(path, "wb");
    if (!f) return 0;
    fprintf(f, "P6\n%i %i\n255\n", W, H);
    unsigned char *buf = (unsigned char *)malloc((size_t)W * H * 3);
    or_to_bytes(rgb, W * H * 3, buf);
    size_t t = fwrite(buf, (size_t)W * H * 3, 1, f);
    free(buf);
    fclose(f);
    return t == 1;
}

/* imageWriter.h:342-404; depthN already normalised once */
int or_write_depth_bmp(const char *path, const float *rgb_unused, const float *depthN, int W, int H)
{
    (void)rgb_unused;
    FILE *f = fopen(path, "wb");
    if (!f) return 0;
    unsigned char fh[14], ih[40];
    bmp_headers(fh, ih, (unsigned)(3 * W * H), W, H);
    fwrite(fh, 1, 14, f), fwrite(ih, 1, 40, f);
    int rowsize = ((24 * W + 31) / 32) * 4;
    /* the reference indexes a rowsize*H buffer at up to 9*W per row; give it room */
    size_t cap = (size_t)rowsize * H + (size_t)9 * W + 16;
    unsigned char *buf = (unsigned char *)calloc(cap, 1);
    for (int y = 0; y < H; ++y) {
        int i = 0;
        for (; i < W * 3; ++i) {
            if ((size_t)y * W + i < (size_t)W * H) {
                unsigned char v = to_byte(1 - depthN[y * W + i]);
                buf[y * rowsize + i * 3] = buf[y * rowsize + i * 3 + 1] = buf[y * rowsize + i * 3 + 2] = v;
            }
        }
        for (; i < rowsize; ++i) buf[y * rowsize + i] = 1;
    }
    size_t t = fwrite(buf, (size_t)W * H * 3, 1, f);
    free(buf);
    fclose(f);
    return t == 1;
}

/* imageWriter.h:406-470; blur is (W-12)*(H-12)*3.  The writers share the member buffer
 * imageC (imageWriter.h:239) and the padding loop below writes three bytes early, so when
 * rows need padding the last three bytes of every row keep what the previous writer left:
 * `prev` (may be NULL = zeros) is that content, n_prev bytes - on the 'r' path the pixel
 * bytes of depth.bmp (main.cpp:375-377). */
int or_write_blurred_bmp2(const char *path, const float *blur, int W, int H, const unsigned char *prev,
                          size_t n_prev)
{
    FILE *f = fopen(path, "wb");
    if (!f) return 0;
    int bw = W - 12, bh = H - 12;
    unsigned char fh[14], ih[40];
    bmp_headers(fh, ih, (unsigned)(bw * bh), bw, bh);
    fwrite(fh, 1, 14, f), fwrite(ih, 1, 40, f);
    int rowsize = ((24 * bw + 31) / 32) * 4;
    unsigned char *buf = (unsigned char *)calloc((size_t)rowsize * bh + 8, 1);
    if (prev) memcpy(buf, prev, n_prev < (size_t)rowsize * bh ? n_prev : (size_t)rowsize * bh);
    for (int y = 0; y < bh; ++y) {
        int i = 0;
        for (; i < bw * 3; i += 3) {
            buf[y * rowsize + i] = to_byte(blur[y * bw * 3 + i + 2]);
            buf[y * rowsize + i + 1] = to_byte(blur[y * bw * 3 + i + 1]);
            buf[y * rowsize + i + 2] = to_byte(blur[y * bw * 3 + i]);
        }
        for (; i < rowsize; ++i) buf[y * rowsize + i - 3] = 1;
    }
    size_t t = fwrite(buf, (size_t)bw * bh * 3, 1, f);
    free(buf);
    fclose(f);
    return t == 1;
}

int or_write_blurred_bmp(const char *path, const float *blur, int W, int H)
{
    return or_write_blurred_bmp2(path, blur, W, H, NULL, 0);
}

/* imageWriter.h:472-488: the per-row update of an existing result.bmp */
int or_update_bmp_rows(const char *path, const float *rgb, int W, int x, int y, int w, int h)
{
    FILE *f = fopen(path, "r+b");
    if (!f) return 0;
    unsigned char *row = (unsigned char *)malloc((size_t)w * 3);
    for (int Y = y; Y < y + h; ++Y) {
        for (int j = x; j < x + w; ++j) {
            row[(j - x) * 3] = to_byte(rgb[(size_t)Y * W * 3 + j * 3 + 2]);
            row[(j - x) * 3 + 1] = to_byte(rgb[(size_t)Y * W * 3 + j * 3 + 1]);
            row[(j - x) * 3 + 2] = to_byte(rgb[(size_t)Y * W * 3 + j * 3]);
        }
        fseek(f, 54 + ((long)Y * w + x) * 3, SEEK_SET);
        fwrite(row, 1, (size_t)w * 3, f);
    }
    free(row);
    fclose(f);
    return 1;
}
