// scene_loader.cpp - Wavefront OBJ/MTL reader producing the flat Mesh arrays.
// Behaviour follows the reference loader (mesh.cpp:96-480, mesh.h:62-135),
// including its quirks, because they decide which material a triangle gets:
//   * one Material object is reused for every "newmtl" block and only its
//     "is set" flags are cleared between blocks (mesh.cpp:368-393), so values a
//     block omits (Ni, illum, Ns, the texture name ...) carry over from the
//     previous block;
//   * a block ends at a blank line or at end of file (mesh.cpp:383-397, 465-473);
//   * material 0 is a built-in grey one (mesh.cpp:110-119);
//   * polygons are fanned (0, i+1, i+2) (mesh.cpp:303-325);
//   * lines are read through a 256-byte buffer (mesh.cpp:26,153).
// Textures are read from the PNG the material names (png_decoder.cpp) or, when that file
// is not there, from a "<name>.rtex" file next to the scene (pre-decoded
// RGBA8, zlib-compressed; see tools/decode_textures.py); '\' in a map_Kd path
// becomes '/', relative paths resolve against the .obj directory and absolute
// paths are not found, as on any machine but the scene authors'.
// Progress and warning lines the reference prints go to stderr here: stdout is left to
// the embedding program (bench.py prints one JSON line there).
#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "rt_host.hpp"

namespace rtamd {

namespace {

constexpr int kLineLen = 256;

struct MtlState { // the carried-over Material of mesh.cpp:368
    float       Kd[3] = {0, 0, 0}, Ks[3] = {0, 0, 0};
    float       Ns = 0, Ni = 0;
    int         illum = 0;
    bool        hasKd = false, hasKa = false, hasKs = false, hasNi = false, hasTr = false;
    std::string texture;
    void        clearFlags() { hasKd = hasKa = hasKs = hasNi = hasTr = false; }
    bool        valid() const { return hasKd || hasKa || hasKs || hasTr; }
};

struct PendingMaterial {
    RtMaterial  m;
    std::string name, texture;
};

const char* skipSpace(const char* p)
{
    while (isspace((unsigned char)*p)) ++p;
    return p;
}

std::string firstToken(const char* p)
{
    p = skipSpace(p);
    const char* e = p;
    while (*e && !isspace((unsigned char)*e)) ++e;
    return std::string(p, e);
}

void readMtl(const std::string& file, std::vector<PendingMaterial>& out,
             std::map<std::string, int>& index)
{
    FILE* in = fopen(file.c_str(), "r");
    if (!in) {
        fprintf(stderr, "  Warning! Material file '%s' not found!\n", file.c_str());
        return;
    }
    MtlState    cur;
    std::string key;
    bool        indef = false;
    auto        commit = [&]() {
        if (index.find(key) != index.end()) return;
        PendingMaterial pm;
        memset(&pm.m, 0, sizeof pm.m);
        memcpy(pm.m.Kd, cur.Kd, sizeof cur.Kd);
        memcpy(pm.m.Ks, cur.Ks, sizeof cur.Ks);
        pm.m.Ns = cur.Ns, pm.m.Ni = cur.Ni, pm.m.illum = cur.illum;
        pm.m.has_Ni = cur.hasNi ? 1 : 0;
        pm.m.texture = -1;
        pm.name = key;
        pm.texture = cur.texture;
        index[key] = (int)out.size() + 1; // material 0 is the built-in one
        out.push_back(pm);
    };
    char line[kLineLen];
    for (;;) {
        memset(line, 0, sizeof line);
        const bool ateof_before = feof(in);
        if (ateof_before) break;
        if (!fgets(line, kLineLen, in)) line[0] = '\0';
        float f1 = 0, f2 = 0, f3 = 0;
        if (line[0] == '#') continue;
        if (line[0] == '\0' || isspace((unsigned char)line[0])) {
            if (indef && !key.empty() && cur.valid()) {
                commit();
                cur.clearFlags();
            }
            if (line[0] == '\0') break;
        } else if (!strncmp(line, "newmtl ", 7)) {
            key = firstToken(line + 7);
            indef = true;
        } else if (!strncmp(line, "Kd ", 3)) {
            sscanf(line + 3, "%f %f %f", &f1, &f2, &f3);
            cur.Kd[0] = f1, cur.Kd[1] = f2, cur.Kd[2] = f3, cur.hasKd = true;
        } else if (!strncmp(line, "Ka ", 3)) {
            cur.hasKa = true;
        } else if (!strncmp(line, "Ks ", 3)) {
            sscanf(line + 3, "%f %f %f", &f1, &f2, &f3);
            cur.Ks[0] = f1, cur.Ks[1] = f2, cur.Ks[2] = f3, cur.hasKs = true;
        } else if (!strncmp(line, "Ns ", 3)) {
            sscanf(line + 3, "%f", &f1);
            cur.Ns = f1;
        } else if (!strncmp(line, "Ni ", 3)) {
            sscanf(line + 3, "%f", &f1);
            cur.Ni = f1, cur.hasNi = true;
        } else if (!strncmp(line, "illum ", 6)) {
            int v = -1;
            sscanf(line + 6, "%i", &v);
            cur.illum = v;
        } else if (!strncmp(line, "map_Kd ", 7)) {
            std::string t = line + 7;
            if (!t.empty() && t.back() == '\n') t.pop_back();
            cur.texture = t;
        } else if (!strncmp(line, "Tr ", 3) || !strncmp(line, "d ", 2)) {
            cur.hasTr = true;
        }
        if (feof(in) && indef && cur.valid() && !key.empty()) commit();
    }
    fclose(in);
    fprintf(stderr, "%u  materials loaded.\n", (unsigned)out.size() + 1);
}

bool readRtex(const std::string& path, Texture& t)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char hdr[16];
    bool          ok = fread(hdr, 1, 16, f) == 16 && !memcmp(hdr, "RTEX", 4);
    uint32_t      w = 0, h = 0, zl = 0;
    if (ok) {
        memcpy(&w, hdr + 4, 4), memcpy(&h, hdr + 8, 4), memcpy(&zl, hdr + 12, 4);
        ok = w > 0 && h > 0 && w <= 16384 && h <= 16384;
    }
    std::vector<unsigned char> z;
    if (ok) { // the compressed length the header states has to fit the file
        const long at = ftell(f);
        ok = fseek(f, 0, SEEK_END) == 0 && ftell(f) >= at && (unsigned long)(ftell(f) - at) >= zl &&
             fseek(f, at, SEEK_SET) == 0;
    }
    if (ok) {
        z.resize(zl);
        ok = fread(z.data(), 1, zl, f) == zl;
    }
    fclose(f);
    if (!ok) return false;
    t.rgba.resize((size_t)w * h * 4);
    uLongf n = (uLongf)t.rgba.size();
    if (uncompress(t.rgba.data(), &n, z.data(), zl) != Z_OK || n != t.rgba.size()) return false;
    t.width = (int)w, t.height = (int)h, t.path = path;
    return true;
}

} // namespace

bool Mesh::loadMesh(const char* filename, bool /*randomizeTriangulation*/, bool loadTextures)
{
    *this = Mesh();
    FILE* in = fopen(filename, "r");
    if (!in) return false;

    RtMaterial grey; // mesh.cpp:110-119
    memset(&grey, 0, sizeof grey);
    grey.Kd[0] = grey.Kd[1] = grey.Kd[2] = 0.5f;
    grey.Ks[0] = grey.Ks[1] = grey.Ks[2] = 0.5f;
    grey.Ns = 96.7f, grey.illum = 2, grey.texture = -1;
    materials.push_back(grey);
    material_names.push_back("StandardMaterialInitFromTriMesh");

    std::string dir(filename);
    for (auto& c : dir)
        if (c == '\\') c = '/';
    const size_t slash = dir.rfind('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);

    std::vector<PendingMaterial> file_mats;
    std::map<std::string, int>   mat_index;
    int                          current = 0; // material of the faces being read; 0 when unknown
    std::vector<int>             vi, ti;
    char                         line[kLineLen];
    for (;;) {
        memset(line, 0, sizeof line);
        if (feof(in) || !fgets(line, kLineLen, in)) break;
        if (line[0] == '#' || line[0] == '\0' || isspace((unsigned char)line[0])) continue;
        if (!strncmp(line, "mtllib ", 7)) {
            const char* p = skipSpace(line + 7);
            size_t      n = 0;
            while (p[n] && (unsigned char)p[n] >= 32 && (unsigned char)p[n] != 255) ++n;
            readMtl(dir + std::string(p, n), file_mats, mat_index);
        } else if (!strncmp(line, "usemtl ", 7)) {
            auto it = mat_index.find(firstToken(line + 7));
            current = it == mat_index.end() ? 0 : it->second;
        } else if (!strncmp(line, "v ", 2)) {
            float x = 0, y = 0, z = 0;
            sscanf(line + 2, "%f %f %f", &x, &y, &z);
            positions.insert(positions.end(), {x, y, z});
        } else if (!strncmp(line, "vt ", 3)) {
            float u = 0, v = 0;
            sscanf(line + 3, "%f %f", &u, &v);
            texcoords.insert(texcoords.end(), {u, v, 0.0f});
        } else if (!strncmp(line, "f ", 2)) {
            // corners are "v", "v/t", "v//n" or "v/t/n"; only v and t are kept.  A corner
            // without t contributes nothing to the t list, which is then padded with
            // index 0 (mesh.cpp:298-299)
            vi.clear(), ti.clear();
            const char* p = line + 2;
            for (;;) {
                while (*p == ' ') ++p;
                if (*p == '\0' || *p == '\n' || *p == '\r') break;
                int field = 0;
                for (;;) {
                    const char* q = p;
                    while (*q && *q != '/' && *q != ' ' && *q != '\n' && *q != '\r') ++q;
                    if (q != p) {
                        // 1-based; a negative index counts back from the entries read so far
                        const int raw = atoi(std::string(p, q).c_str());
                        if (field == 0) vi.push_back(raw < 0 ? (int)vertexCount() + raw : raw - 1);
                        else if (field == 1) ti.push_back(raw < 0 ? (int)(texcoords.size() / 3) + raw : raw - 1);
                    }
                    ++field;
                    p = q;
                    if (*p != '/') break;
                    ++p;
                }
            }
            if (ti.size() != vi.size()) ti.resize(vi.size(), 0);
            for (size_t i = 0; i + 2 < vi.size(); ++i) {
                tri_vertex.insert(tri_vertex.end(), {vi[0], vi[i + 1], vi[i + 2]});
                tri_texcoord.insert(tri_texcoord.end(), {ti[0], ti[i + 1], ti[i + 2]});
                tri_material.push_back(current);
            }
            if (vi.size() < 3) fprintf(stderr, "TriMesh::LOAD: Unexpected number of face vertices (<3). Ignoring face");
        }
    }
    fclose(in);

    // The reference indexes its vertex and texcoord arrays with whatever the file says
    // (mesh.cpp:298-325).  Here a face that names a vertex the file does not have is dropped
    // and a texcoord index outside the list reads entry 0, before anything indexes with them.
    {
        const int nv = (int)vertexCount(), ntc = (int)(texcoords.size() / 3);
        size_t    kept = 0, dropped = 0;
        for (size_t t = 0; t < tri_material.size(); ++t) {
            bool ok = true;
            for (int j = 0; j < 3; ++j) ok = ok && tri_vertex[3 * t + j] >= 0 && tri_vertex[3 * t + j] < nv;
            if (!ok) {
                ++dropped;
                continue;
            }
            for (int j = 0; j < 3; ++j) {
                const int tc = tri_texcoord[3 * t + j];
                tri_vertex[3 * kept + j] = tri_vertex[3 * t + j];
                tri_texcoord[3 * kept + j] = (tc >= 0 && tc < ntc) ? tc : 0;
            }
            tri_material[kept++] = tri_material[t];
        }
        tri_vertex.resize(3 * kept), tri_texcoord.resize(3 * kept), tri_material.resize(kept);
        if (dropped)
            fprintf(stderr, "TriMesh::LOAD: %zu faces name a vertex outside 1..%d. Ignoring them\n", dropped, nv);
    }

    // every material tries to open its (possibly inherited) texture, mesh.cpp:346-348
    for (auto& pm : file_mats) {
        if (loadTextures && !pm.texture.empty()) {
            std::string rel = pm.texture;
            for (auto& c : rel)
                if (c == '\\') c = '/';
            while (!rel.empty() && (rel.back() == '\r' || rel.back() == '\n')) rel.pop_back();
            const bool absolute =
                !rel.empty() && (rel[0] == '/' || (rel.size() > 1 && isalpha((unsigned char)rel[0]) && rel[1] == ':'));
            if (!rel.empty() && !absolute) {
                // the image file itself if it is there (mesh.h:120), else its pre-decoded copy
                const std::string png = dir + rel, path = dir + rel + ".rtex";
                int               found = -1;
                for (size_t i = 0; i < textures.size(); ++i)
                    if (textures[i].path == path || textures[i].path == png) found = (int)i;
                if (found < 0) {
                    Texture     t;
                    std::string why;
                    if (readPng(png, t, why) || readRtex(path, t)) {
                        textures.push_back(std::move(t));
                        found = (int)textures.size() - 1;
                    } else {
                        fprintf(stderr, "failed to read texture: %s\n", pm.texture.c_str());
                    }
                }
                pm.m.texture = found;
            }
        }
        materials.push_back(pm.m);
        material_names.push_back(pm.name);
    }
    normals.assign(positions.size(), 0.0f);
    return true;
}

// mesh.cpp:32-52: unweighted sum of unit face normals per vertex, normalised.
// normalize() there multiplies by 1/length and leaves a zero vector alone
// (Vec3D.h:142-151).
void Mesh::computeVertexNormals()
{
    normals.assign(positions.size(), 0.0f);
    auto unit = [](float v[3]) {
        const float len = (float)std::sqrt((double)(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
        if (len == 0.0f) return;
        const float r = 1.0f / len;
        v[0] *= r, v[1] *= r, v[2] *= r;
    };
    for (size_t t = 0; t < triangleCount(); ++t) {
        const float* a = &positions[3 * tri_vertex[3 * t]];
        const float* b = &positions[3 * tri_vertex[3 * t + 1]];
        const float* c = &positions[3 * tri_vertex[3 * t + 2]];
        const float  ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
        const float  ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        float n[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2],
                      ab[0] * ac[1] - ab[1] * ac[0]};
        unit(n);
        for (int j = 0; j < 3; ++j) {
            float* d = &normals[3 * tri_vertex[3 * t + j]];
            d[0] += n[0], d[1] += n[1], d[2] += n[2];
        }
    }
    for (size_t v = 0; v < vertexCount(); ++v) unit(&normals[3 * v]);
}

} // namespace rtamd
