// scene_obj.cpp -- Wavefront OBJ (+ MTL, factor-only materials) ingestion: produces the host arrays that the
// reference's Scene::Load (tools/sceneBuilder.cpp:27-258) builds from an assimp import -- the concatenated
// vertex / index / uv / normal streams with one MeshInfo per mesh, the Material table and one instance per mesh --
// without assimp (an un-vendored submodule of the reference, absent offline).  SURVEY.md 8(f) rank 2.
//
// What corresponds to what:
//   * mesh split: a new mesh starts at every `o` / `g` / `usemtl` change that is followed by faces (assimp's OBJ
//     importer makes one aiMesh per object and material);
//   * aiProcess_Triangulate: polygons are fanned from their first vertex;
//   * aiProcess_JoinIdenticalVertices: one vertex per distinct (v, vt, vn) index triple of a mesh, in first-use
//     order; indices are mesh-local (sceneBuilder.cpp:69-101 concatenates the meshes, MeshInfo carries the offsets
//     in floats / indices: `size() * 3`, :73-76);
//   * aiProcess_GenSmoothNormals: a mesh with a face vertex that has no `vn` gets smooth normals -- the
//     un-normalised face normals (area weighted) summed over the faces sharing the POSITION, normalised.  assimp's
//     own smoothing-angle and epsilon rules are not reproduced: parity unpinned at this boundary;
//   * uv: (u, v, 0) like aiVector3D texture coordinates, zero when absent (:86-87);
//   * materials (sceneBuilder.cpp:103-193, factor branches only -- the texture branches need embedded glTF images
//     and the live shader stubs every texture fetch, shader.cl:379-445): Kd + d -> albedo rgba, Pm -> metallic
//     (default 0), Pr -> roughness (without Pr: 1 - sqrt(Ns / 1000), the inverse of Blender's exporter; else 0.5),
//     Tf (mean) -> transmission (default 0, :171-175), Ni -> ior (default 1.45, :177-181), all *TexIdx = -1;
//     a mesh without `usemtl` gets a 0.6-grey default material appended last (assimp's DefaultMaterial);
//   * instances: OBJ has no node hierarchy, so every mesh is instanced once with the identity transform,
//     SBTOffset 0 and customInstanceID = its material index (BuildInstance, sceneBuilder.cpp:287-315).
#include "../../include/rdx.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include <cstdarg>

namespace rdx {
int fail_text(const char* text);                 // rdx_runtime.cpp: sets rdx_last_error(), returns -1
static int fail(const char* fmt, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    return fail_text(buf);
}
}

namespace {

struct Mtl { std::string name; rdx_material m; };

void default_material(rdx_material& m)
{
    std::memset(&m, 0, sizeof m);
    m.albedo[0] = m.albedo[1] = m.albedo[2] = 0.6f; m.albedo[3] = 1.0f;
    m.metallic = 0.0f; m.roughness = 0.5f; m.transmission = 0.0f; m.ior = 1.45f;
    m.albedoTexIdx = m.metallicTexIdx = m.roughnessTexIdx = m.normalTexIdx = -1;
}

std::string dir_of(const std::string& path)
{
    const size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

// reads one logical line (joins trailing-backslash continuations); false at end of file
bool read_line(FILE* fp, std::string& out)
{
    out.clear();
    char buf[4096];
    bool any = false;
    while (fgets(buf, sizeof buf, fp)) {
        any = true;
        out += buf;
        while (!out.empty() && (out.back() == '\n' || out.back() == '\r')) out.pop_back();
        if (!out.empty() && out.back() == '\\') { out.pop_back(); continue; }
        if (std::strlen(buf) == sizeof buf - 1 && buf[sizeof buf - 2] != '\n') continue;   // long line: keep reading
        break;
    }
    return any;
}

bool load_mtl(const std::string& path, std::vector<Mtl>& out)
{
    FILE* fp = std::fopen(path.c_str(), "r");
    if (!fp) return false;
    std::string line;
    Mtl* cur = nullptr;
    bool haveRough = false;
    float ns = -1.0f;
    auto finish = [&]() {
        if (cur && !haveRough && ns >= 0.0f) {          // Blender's exporter: Ns = (1 - roughness)^2 * 1000
            float r = 1.0f - std::sqrt(std::fmin(std::fmax(ns / 1000.0f, 0.0f), 1.0f));
            cur->m.roughness = r;
        }
    };
    while (read_line(fp, line)) {
        char key[64];
        int used = 0;
        if (std::sscanf(line.c_str(), " %63s%n", key, &used) != 1 || key[0] == '#') continue;
        const char* rest = line.c_str() + used;
        if (!std::strcmp(key, "newmtl")) {
            finish();
            out.emplace_back();
            cur = &out.back();
            default_material(cur->m);
            char name[1024] = "";
            std::sscanf(rest, " %1023[^\n]", name);
            cur->name = name;
            haveRough = false; ns = -1.0f;
            continue;
        }
        if (!cur) continue;
        float a, b, c;
        if (!std::strcmp(key, "Kd") && std::sscanf(rest, "%f %f %f", &a, &b, &c) == 3) { cur->m.albedo[0] = a; cur->m.albedo[1] = b; cur->m.albedo[2] = c; }
        else if (!std::strcmp(key, "d") && std::sscanf(rest, "%f", &a) == 1) cur->m.albedo[3] = a;
        else if (!std::strcmp(key, "Tr") && std::sscanf(rest, "%f", &a) == 1) cur->m.albedo[3] = 1.0f - a;
        else if (!std::strcmp(key, "Pm") && std::sscanf(rest, "%f", &a) == 1) cur->m.metallic = a;
        else if (!std::strcmp(key, "Pr") && std::sscanf(rest, "%f", &a) == 1) { cur->m.roughness = a; haveRough = true; }
        else if (!std::strcmp(key, "Ns") && std::sscanf(rest, "%f", &a) == 1) ns = a;
        else if (!std::strcmp(key, "Ni") && std::sscanf(rest, "%f", &a) == 1) cur->m.ior = a;
        else if (!std::strcmp(key, "Tf")) {
            const int n = std::sscanf(rest, "%f %f %f", &a, &b, &c);
            if (n == 3) cur->m.transmission = (a + b + c) / 3.0f; else if (n >= 1) cur->m.transmission = a;
        }
    }
    finish();
    std::fclose(fp);
    return true;
}

struct MeshBuild {
    int material = -1;                                   // index into the MTL table, -1 = none
    std::map<std::tuple<int, int, int>, uint32_t> remap;  // (v, vt, vn) -> mesh-local vertex
    std::vector<int> vpos;                                // mesh-local vertex -> position index (smooth normals)
    std::vector<float> pos, uv, nrm;                      // 3 floats per vertex
    std::vector<uint32_t> tri;
    bool missingNormal = false;
};

template <class T> T* dup(const std::vector<T>& v)
{
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

} // namespace

extern "C" void rdx_obj_free(rdx_obj_scene* s)
{
    if (!s) return;
    std::free(s->meshInfo); std::free(s->vertex); std::free(s->index); std::free(s->uv); std::free(s->normal);
    std::free(s->materials); std::free(s->meshVertexCount); std::free(s->meshTriangleCount);
    std::memset(s, 0, sizeof *s);
}

extern "C" int rdx_obj_load(const char* path, rdx_obj_scene* out)
{
    if (!path || !out) return rdx::fail("rdx_obj_load: null argument");
    std::memset(out, 0, sizeof *out);
    FILE* fp = std::fopen(path, "r");
    if (!fp) return rdx::fail("rdx_obj_load: cannot open '%s'", path);
    std::vector<float> P, T, N;                           // file-global v / vt / vn
    std::vector<Mtl> mtls;
    std::vector<MeshBuild> meshes;
    int curMtl = -1;
    bool needNew = true;
    std::string line;
    long lineNo = 0;
    while (read_line(fp, line)) {
        ++lineNo;
        const char* s = line.c_str();
        while (*s == ' ' || *s == '\t') ++s;
        if (!*s || *s == '#') continue;
        float a, b, c;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            if (std::sscanf(s + 1, "%f %f %f", &a, &b, &c) != 3) { std::fclose(fp); return rdx::fail("%s:%ld: malformed vertex", path, lineNo); }
            P.push_back(a); P.push_back(b); P.push_back(c);
        } else if (s[0] == 'v' && s[1] == 'n') {
            if (std::sscanf(s + 2, "%f %f %f", &a, &b, &c) != 3) { std::fclose(fp); return rdx::fail("%s:%ld: malformed normal", path, lineNo); }
            N.push_back(a); N.push_back(b); N.push_back(c);
        } else if (s[0] == 'v' && s[1] == 't') {
            const int n = std::sscanf(s + 2, "%f %f", &a, &b);
            if (n < 1) { std::fclose(fp); return rdx::fail("%s:%ld: malformed texture coordinate", path, lineNo); }
            T.push_back(a); T.push_back(n >= 2 ? b : 0.0f);
        } else if ((s[0] == 'o' || s[0] == 'g') && (s[1] == ' ' || s[1] == '\t' || !s[1])) {
            needNew = true;
        } else if (!std::strncmp(s, "usemtl", 6)) {
            char name[1024] = "";
            std::sscanf(s + 6, " %1023[^\n]", name);
            curMtl = -1;
            for (size_t i = 0; i < mtls.size(); ++i) if (mtls[i].name == name) curMtl = (int)i;
            if (curMtl < 0) { std::fclose(fp); return rdx::fail("%s:%ld: material '%s' is not defined by any mtllib", path, lineNo, name); }
            needNew = true;
        } else if (!std::strncmp(s, "mtllib", 6)) {
            char name[1024] = "";
            std::sscanf(s + 6, " %1023[^\n]", name);
            if (!load_mtl(dir_of(path) + name, mtls)) { std::fclose(fp); return rdx::fail("%s:%ld: cannot open material library '%s'", path, lineNo, name); }
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            if (needNew || meshes.empty()) { meshes.emplace_back(); meshes.back().material = curMtl; needNew = false; }
            MeshBuild& M = meshes.back();
            std::vector<uint32_t> poly;
            const char* q = s + 1;
            for (;;) {
                while (*q == ' ' || *q == '\t') ++q;
                if (!*q) break;
                int vi = 0, ti = 0, ni = 0;
                char* e;
                vi = (int)std::strtol(q, &e, 10);
                if (e == q) { std::fclose(fp); return rdx::fail("%s:%ld: malformed face", path, lineNo); }
                q = e;
                if (*q == '/') { ++q; if (*q != '/') { ti = (int)std::strtol(q, &e, 10); q = e; } if (*q == '/') { ++q; ni = (int)std::strtol(q, &e, 10); q = e; } }
                const int nP = (int)(P.size() / 3), nT = (int)(T.size() / 2), nN = (int)(N.size() / 3);
                if (vi < 0) vi = nP + 1 + vi;
                if (ti < 0) ti = nT + 1 + ti;
                if (ni < 0) ni = nN + 1 + ni;
                if (vi < 1 || vi > nP || ti < 0 || ti > nT || ni < 0 || ni > nN) { std::fclose(fp); return rdx::fail("%s:%ld: face index out of range", path, lineNo); }
                auto key = std::make_tuple(vi, ti, ni);
                auto it = M.remap.find(key);
                if (it == M.remap.end()) {
                    const uint32_t id = (uint32_t)M.vpos.size();
                    it = M.remap.emplace(key, id).first;
                    M.vpos.push_back(vi - 1);
                    for (int k = 0; k < 3; ++k) M.pos.push_back(P[3 * (vi - 1) + k]);
                    M.uv.push_back(ti ? T[2 * (ti - 1)] : 0.0f); M.uv.push_back(ti ? T[2 * (ti - 1) + 1] : 0.0f); M.uv.push_back(0.0f);
                    if (ni) for (int k = 0; k < 3; ++k) M.nrm.push_back(N[3 * (ni - 1) + k]);
                    else { M.nrm.push_back(0.f); M.nrm.push_back(0.f); M.nrm.push_back(0.f); M.missingNormal = true; }
                }
                poly.push_back(it->second);
            }
            if (poly.size() < 3) { std::fclose(fp); return rdx::fail("%s:%ld: face with fewer than 3 vertices", path, lineNo); }
            for (size_t k = 1; k + 1 < poly.size(); ++k) { M.tri.push_back(poly[0]); M.tri.push_back(poly[k]); M.tri.push_back(poly[k + 1]); }
        }
        // everything else (s, l, p, vp, curves ...) is ignored, as aiProcess_SortByPType + the triangle-only loop do
    }
    std::fclose(fp);
    // drop face-less meshes, generate smooth normals where needed
    std::vector<MeshBuild> keep;
    for (auto& M : meshes) if (!M.tri.empty()) keep.push_back(std::move(M));
    if (keep.empty()) return rdx::fail("rdx_obj_load: '%s' contains no faces", path);
    bool needDefault = false;
    for (auto& M : keep) {
        if (M.material < 0) needDefault = true;
        if (!M.missingNormal) continue;
        std::map<int, std::tuple<double, double, double>> acc;      // position index -> summed face normal
        for (size_t t = 0; t + 2 < M.tri.size(); t += 3) {
            const float* p0 = &M.pos[3 * M.tri[t]]; const float* p1 = &M.pos[3 * M.tri[t + 1]]; const float* p2 = &M.pos[3 * M.tri[t + 2]];
            const double e1[3] = {(double)p1[0] - p0[0], (double)p1[1] - p0[1], (double)p1[2] - p0[2]};
            const double e2[3] = {(double)p2[0] - p0[0], (double)p2[1] - p0[1], (double)p2[2] - p0[2]};
            const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            for (int k = 0; k < 3; ++k) {
                auto& a = acc[M.vpos[M.tri[t + k]]];
                std::get<0>(a) += n[0]; std::get<1>(a) += n[1]; std::get<2>(a) += n[2];
            }
        }
        for (size_t v = 0; v < M.vpos.size(); ++v) {
            const auto& a = acc[M.vpos[v]];
            const double l = std::sqrt(std::get<0>(a) * std::get<0>(a) + std::get<1>(a) * std::get<1>(a) + std::get<2>(a) * std::get<2>(a));
            M.nrm[3 * v] = l > 0 ? (float)(std::get<0>(a) / l) : 0.0f;
            M.nrm[3 * v + 1] = l > 0 ? (float)(std::get<1>(a) / l) : 1.0f;
            M.nrm[3 * v + 2] = l > 0 ? (float)(std::get<2>(a) / l) : 0.0f;
        }
    }
    std::vector<rdx_material> mats;
    for (auto& m : mtls) mats.push_back(m.m);
    if (needDefault) { rdx_material d; default_material(d); mats.push_back(d); }
    std::vector<rdx_mesh_info> info;
    std::vector<float> vertex, uv, normal;
    std::vector<uint32_t> index, vcount, tcount;
    for (auto& M : keep) {
        rdx_mesh_info mi;
        std::memset(&mi, 0, sizeof mi);
        mi.vertexOffset = (int32_t)vertex.size();           // offsets in floats / indices (sceneBuilder.cpp:73-76)
        mi.indexOffset = (int32_t)index.size();
        mi.uvOffset = (int32_t)uv.size();
        mi.normalOffset = (int32_t)normal.size();
        mi.materialIndex = M.material >= 0 ? M.material : (int32_t)mats.size() - 1;
        info.push_back(mi);
        vertex.insert(vertex.end(), M.pos.begin(), M.pos.end());
        uv.insert(uv.end(), M.uv.begin(), M.uv.end());
        normal.insert(normal.end(), M.nrm.begin(), M.nrm.end());
        index.insert(index.end(), M.tri.begin(), M.tri.end());
        vcount.push_back((uint32_t)M.vpos.size());
        tcount.push_back((uint32_t)(M.tri.size() / 3));
    }
    out->nmeshes = (uint32_t)keep.size();
    out->nvertices = (uint32_t)(vertex.size() / 3);
    out->ntriangles = (uint32_t)(index.size() / 3);
    out->nmaterials = (uint32_t)mats.size();
    out->meshInfo = dup(info); out->vertex = dup(vertex); out->index = dup(index); out->uv = dup(uv); out->normal = dup(normal);
    out->materials = dup(mats); out->meshVertexCount = dup(vcount); out->meshTriangleCount = dup(tcount);
    if (!out->meshInfo || !out->vertex || !out->index || !out->uv || !out->normal || !out->materials || !out->meshVertexCount ||
        !out->meshTriangleCount) { rdx_obj_free(out); return rdx::fail("rdx_obj_load: out of memory"); }
    return 0;
}
