// novum_host.cpp — the host side the reference keeps around the hot path, restated in plain
// C++ (the reference's lives in CUDA-typed .cu/.cuh files): `.rendertron` parser
// (objects.cuh:844-943), OBJ reader (main.cu:936-1068), material table (main.cu:397-467),
// SAH BVH builder (main.cu:20-233), camera set-up (objects.cuh:221-264, 309-325), finalise
// (main.cu:860-870), BMP output (imageUtil.cu:69-100, 202-232) and initRender (main.cu:235-923)
// for the two unidirectional integrators. It produces the reference's own data model
// (pt_triangle / pt_bvh_node / pt_material ...) and hands it to pt_scene_create.
//
// Behaviour the reference leaves undefined is resolved as SURVEY.md Appendix D lists:
// materials are zero-initialised before their factory runs; faces without `vn` get their
// geometric normal, faces without `vt` get uv (0,0); the std::nth_element fallback of the
// builder is a full sort by (centroid, index); parseVec3's .w is 0.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pt_api.h"

namespace {

struct F3 { float x, y, z; };
inline pt_float4 P4(float x, float y, float z) { return pt_float4{x, y, z, 0.0f}; }
inline float comp(const pt_float4& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

struct MeshLine { std::string path; float mult = 0.0f; float rgb[3] = {0, 0, 0}; int material = 0; };

struct Config {                                   // RenderConfig, objects.cuh:801-842 (hot-path keys)
    int width = 0, height = 0, spp = 0, maxDepth = 0, leafSize = 0;
    std::string name, integrator;
    bool postProcess = false, pinhole = false;
    float camPos[3] = {0, 0, 0}, camRot[3] = {0, 0, 0};
    float fov = 0, aperture = 0, focalDist = 0;
    std::vector<MeshLine> meshes;
};

std::string strip(const std::string& s) {         // trim, util.cuh:288-293
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return s;
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}
bool truthy(std::string v) { for (char& c : v) c = (char)tolower((unsigned char)c); return v == "true"; }   // parseBool, util.cuh:302-306
void vec3_of(const std::string& s, float out[3]) {                                                          // parseVec3, util.cuh:295-300
    // operator>> semantics: stop at the first token that is not a number
    const char* p = s.c_str();
    for (int i = 0; i < 3; i++) {
        char* e = nullptr;
        float v = strtof(p, &e);
        if (e == p) return;
        out[i] = v; p = e;
    }
}

bool parse_config(const std::string& path, Config& c) {            // loadConfig, objects.cuh:844-943
    std::ifstream f(path);
    if (!f.is_open()) { fprintf(stderr, "Error: Could not open config file: %s\n", path.c_str()); return false; }
    std::string raw;
    bool meshes = false;
    while (std::getline(f, raw)) {
        std::string line = strip(raw);
        if (line.empty()) continue;
        if (line.compare(0, 6, "Meshes") == 0) { meshes = true; continue; }
        if (meshes) {
            // path; mult * (r, g, b); materialID
            MeshLine m;
            size_t s1 = line.find(';');
            m.path = strip(line.substr(0, s1));
            if (s1 != std::string::npos) {
                size_t s2 = line.find(';', s1 + 1);
                std::string em = strip(line.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1));
                size_t star = em.find('*'), lp = em.find('('), rp = em.find(')');
                if (star != std::string::npos && lp != std::string::npos) {
                    m.mult = std::stof(em.substr(0, star));
                    std::string v = em.substr(lp + 1, rp - lp - 1);
                    std::replace(v.begin(), v.end(), ',', ' ');
                    vec3_of(v, m.rgb);
                }
                if (s2 != std::string::npos) {
                    size_t s3 = line.find(';', s2 + 1);
                    m.material = std::stoi(strip(line.substr(s2 + 1, s3 == std::string::npos ? std::string::npos : s3 - s2 - 1)));
                }
            }
            c.meshes.push_back(m);
            continue;
        }
        size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        std::string key = strip(line.substr(0, colon)), val = strip(line.substr(colon + 1));
        if (val.empty()) continue;
        if (key == "width") c.width = std::stoi(val);
        else if (key == "height") c.height = std::stoi(val);
        else if (key == "Integrator") c.integrator = val;
        else if (key == "Name") c.name = val;
        else if (key == "Sample Count") c.spp = std::stoi(val);
        else if (key == "Unidirectional Max Depth") c.maxDepth = std::stoi(val);
        else if (key == "BVH recommended leaf size") c.leafSize = std::stoi(val);
        else if (key == "Pinhole Camera") c.pinhole = truthy(val);
        else if (key == "Post Process") c.postProcess = truthy(val);
        else if (key == "Camera Position") vec3_of(val, c.camPos);
        else if (key == "Camera Rotation") vec3_of(val, c.camRot);
        else if (key == "Camera FOV") c.fov = std::stof(val);
        else if (key == "Camera Apeture") c.aperture = std::stof(val);
        else if (key == "Camera FocalDist") c.focalDist = std::stof(val);
        // every other key (BDPT / VCM settings, objects.cuh:916-939) belongs to out-of-scope integrators
    }
    return true;
}

int integrator_id(const std::string& n) {                           // matchIntegrator, objects.cuh:583-593
    if (n == "UNIDIRECTIONAL") return 0;
    if (n == "BIDIRECTIONAL" || n == "BDPT") return 1;
    if (n == "NAIVE_UNIDIRECTIONAL") return 2;
    if (n == "VCM") return 3;
    if (n == "SPPM") return 4;
    return -1;
}

}  // namespace

struct novum_scene {
    Config cfg;
    std::vector<pt_float4> points, normals, textures;
    std::vector<pt_float2> uvs;
    std::vector<pt_triangle> mesh, lights;
    std::vector<pt_bvh_node> bvh;
    std::vector<int32_t> indices;
    std::vector<pt_material> mats;
    pt_camera cam{};
    int largestLeaf = 0, backups = 0, treeDepth = 0;
};

namespace {

// ---- OBJ (main.cu:936-1068) ---------------------------------------------------------------------
struct Cursor {
    const char* p; const char* end;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) p++; }
    bool number(double& v) { ws(); if (p >= end) return false; char* e; v = strtod(p, &e); if (e == p) return false; p = e; return true; }
    bool token(std::string& t) { ws(); if (p >= end) return false; const char* s = p; while (p < end && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) p++; t.assign(s, p); return true; }
};

inline float dot3(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }      // same contract as the kernels
inline F3 cross3(F3 a, F3 b) { return F3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))}; }

void read_obj(const std::string& file, novum_scene& S, const float e[3], int materialID, const float off[3]) {
    std::ifstream in(file, std::ios::binary);
    if (!in.is_open()) { fprintf(stderr, "Error: Could not open OBJ file\n"); return; }
    std::string data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    const int vBase = (int)S.points.size(), nBase = (int)S.normals.size(), tBase = (int)S.uvs.size();
    int nextLight = (int)S.lights.size();
    int zeroUv = -1;
    const bool isLight = dot3(F3{e[0], e[1], e[2]}, F3{e[0], e[1], e[2]}) > 0;
    std::vector<int> vi, ti, ni;
    std::string tok;
    size_t pos = 0;
    while (pos < data.size()) {
        size_t nl = data.find('\n', pos);
        if (nl == std::string::npos) nl = data.size();
        Cursor c{data.data() + pos, data.data() + nl};
        pos = nl + 1;
        if (c.p == c.end || *c.p == '#' || *c.p == 's') continue;
        if (!c.token(tok)) continue;
        if (tok == "v") {
            double x = 0, y = 0, z = 0;
            (void)(c.number(x) && c.number(y) && c.number(z));
            S.points.push_back(P4((float)x + off[0], (float)y + off[1], (float)z + off[2]));
        } else if (tok == "vt") {
            double u = 0, v = 0;
            (void)(c.number(u) && c.number(v));
            S.uvs.push_back(pt_float2{(float)u, (float)(1.0 - v)});
        } else if (tok == "vn") {
            double x = 0, y = 0, z = 0;
            bool ok = c.number(x) && c.number(y) && c.number(z);
            if (!ok || std::isnan(x) || std::isnan(y) || std::isnan(z)) { S.normals.push_back(P4(0, 1, 0)); continue; }
            float fx = (float)x, fy = (float)y, fz = (float)z;
            if (fx * fx + fy * fy + fz * fz < 1e-12f) { fx = 0; fy = 1; fz = 0; }
            S.normals.push_back(P4(fx, fy, fz));
        } else if (tok == "f") {
            vi.clear(); ti.clear(); ni.clear();
            while (c.token(tok)) {
                // v[/vt[/vn]] — an empty field contributes no index (main.cu:1003-1017)
                size_t a = tok.find('/');
                std::string f0 = tok.substr(0, a), f1, f2;
                if (a != std::string::npos) {
                    size_t b = tok.find('/', a + 1);
                    f1 = tok.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1);
                    if (b != std::string::npos) { size_t d = tok.find('/', b + 1); f2 = tok.substr(b + 1, d == std::string::npos ? std::string::npos : d - b - 1); }
                }
                if (!f0.empty()) vi.push_back(std::stoi(f0) - 1);
                if (!f1.empty()) ti.push_back(std::stoi(f1) - 1);
                if (!f2.empty()) ni.push_back(std::stoi(f2) - 1);
            }
            const bool hasUV = ti.size() == vi.size(), hasN = ni.size() == vi.size();
            const int n = (int)vi.size();
            for (int i = 1; i < n - 1; ++i) {                  // fan from the first vertex
                int i0 = vi[0] + vBase, i1 = vi[i] + vBase, i2 = vi[i + 1] + vBase;
                const pt_float4 &p0 = S.points[i0], &p1 = S.points[i1], &p2 = S.points[i2];
                F3 e1{p1.x - p0.x, p1.y - p0.y, p1.z - p0.z}, e2{p2.x - p0.x, p2.y - p0.y, p2.z - p0.z};
                F3 cp = cross3(e1, e2);
                if (dot3(cp, cp) < 1e-18f) continue;           // degenerate-triangle cull, main.cu:1040
                pt_triangle t;
                std::memset(&t, 0, sizeof(t));
                t.aInd = i0; t.bInd = i1; t.cInd = i2;
                if (hasUV) { t.uvaInd = ti[0] + tBase; t.uvbInd = ti[i] + tBase; t.uvcInd = ti[i + 1] + tBase; }
                else {
                    if (zeroUv < 0) { zeroUv = (int)S.uvs.size(); S.uvs.push_back(pt_float2{0.0f, 0.0f}); }
                    t.uvaInd = t.uvbInd = t.uvcInd = zeroUv;
                }
                if (hasN) { t.naInd = ni[0] + nBase; t.nbInd = ni[i] + nBase; t.ncInd = ni[i + 1] + nBase; }
                else {
                    float il = 1.0f / sqrtf(dot3(cp, cp));
                    t.naInd = t.nbInd = t.ncInd = (int)S.normals.size();
                    S.normals.push_back(P4(cp.x * il, cp.y * il, cp.z * il));
                }
                t.materialID = materialID;
                t.emission = P4(e[0], e[1], e[2]);
                t.lightInd = isLight ? nextLight : -51;
                t.triInd = (int)S.mesh.size();
                S.mesh.push_back(t);
                if (isLight) { S.lights.push_back(t); nextLight++; }
            }
        }
    }
}

// ---- materials (objects.cuh:640-791, main.cu:397-467) -------------------------------------------
pt_material base_material() {
    pt_material m;
    std::memset(&m, 0, sizeof(m));
    m.type = PT_MAT_DIFFUSE; m.albedo = P4(0.8f, 0.8f, 0.8f); m.roughness = 0.5f; m.ior = 1.5f; m.specular = 1.0f;
    return m;
}
pt_material lambert(float r, float g, float b) { pt_material m = base_material(); m.albedo = P4(r, g, b); m.roughness = 1.0f; return m; }
pt_material lambert_tex(int start, int w, int h) { pt_material m = base_material(); m.hasTexture = 1; m.startInd = start; m.width = w; m.height = h; m.roughness = 1.0f; return m; }
pt_material conductor(pt_float4 eta, pt_float4 k, float rough) {
    pt_material m = base_material();
    m.type = PT_MAT_METAL; m.eta = eta; m.k = k; m.roughness = rough; m.albedo = P4(1, 1, 1); m.metallic = 1.0f;
    return m;
}
pt_material glassy(float ior, pt_float4 absorb, int priority) {
    pt_material m = base_material();
    m.type = PT_MAT_SMOOTHDIELECTRIC; m.ior = ior; m.albedo = P4(1, 1, 1); m.priority = priority; m.isSpecular = 1; m.boundary = 1; m.absorption = absorb;
    return m;
}
pt_material leafy(int start, int w, int h, float ior, float rough, pt_float4 albedo, float transmission) {
    pt_material m = base_material();
    m.type = PT_MAT_LEAF; m.hasTexture = 1; m.ior = ior; m.roughness = rough; m.albedo = albedo; m.transmission = transmission;
    m.startInd = start; m.width = w; m.height = h; m.thinWalled = 1;
    return m;
}
pt_material mirror_mat() { pt_material m = base_material(); m.type = PT_MAT_DELTAMIRROR; m.isSpecular = 1; return m; }

void material_table(novum_scene& S, const int texStart[4], const int texW[4], const int texH[4]) {
    pt_float4 etaSteel{0.14f, 0.16f, 0.13f, 1.0f}, etaGold = P4(0.17f, 0.35f, 1.5f);
    pt_float4 teaAbs = P4(2.5f * 0.180f, 2.5f * 1.5f, 2.5f * 2.996f);
    pt_float4 leafGreen = P4(0.22f, 0.75f, 0.28f);
    S.mats = {
        glassy(1.0f, P4(0, 0, 0), 99),                               // 0 air
        lambert(0.4f, 0.4f, 0.8f),                                   // 1 blue
        lambert(0.9f, 0.9f, 0.9f),                                   // 2 white
        lambert(0.2f, 0.6f, 0.6f),                                   // 3 teal
        conductor(etaGold, etaGold, 0.05f),                          // 4 gold (k := eta, main.cu:419)
        glassy(1.5f, P4(0, 0, 0), 1),                                // 5 glass
        lambert(0.90f, 0.1f, 0.1f),                                  // 6 red
        conductor(etaSteel, etaSteel, 0.15f),                        // 7 steel
        glassy(1.333f, teaAbs, 2),                                   // 8 tea
        glassy(1.31f, P4(0.2f, 0.2f, 0.2f), 0),                      // 9 ice
        glassy(1.333f, P4(0, 0, 0), 2),                              // 10 water
        lambert_tex(texStart[0], texW[0], texH[0]),                  // 11
        lambert_tex(texStart[1], texW[1], texH[1]),                  // 12
        leafy(texStart[2], texW[2], texH[2], 1.5f, 0.10f, leafGreen, 0.15f),   // 13 leaf
        lambert(0.90f, 0.9f, 0.83f),                                 // 14 stem
        lambert(0.4f, 0.4f, 1.00f),                                  // 15 sky
        leafy(texStart[3], texW[3], texH[3], 1.5f, 0.8f, leafGreen, 0.6f),     // 16 autumn leaf
        lambert(0.8f, 0.8f, 0.8f),                                   // 17 grey
        glassy(2.42f, P4(0, 0, 0), 1),                               // 18 diamond
        mirror_mat(),                                                // 19
        lambert(0.0f, 0.0f, 0.0f),                                   // 20 black
        lambert(0.95f, 0.95f, 0.95f),                                // 21
        lambert(0.5f, 0.5f, 0.5f),                                   // 22
        lambert(0.1f, 0.9f, 0.1f),                                   // 23 green
    };
}

// ---- SAH BVH (main.cu:20-233), iterative, same pre-order numbering ---------------------------------
struct Builder {
    novum_scene& S;
    std::vector<pt_float4> cen, lo, hi;
    int leafMax;
    // fminf / fmaxf with the one freedom IEEE leaves (which zero of -0, +0) fixed as -0 < +0, so that a
    // union over a set does not depend on the order it is taken in (the device builder reduces in parallel).
    static float lo1(float a, float b) { if (a != a) return b; if (b != b) return a; if (a == 0.0f && b == 0.0f) return std::signbit(a) ? a : b; return a < b ? a : b; }
    static float hi1(float a, float b) { if (a != a) return b; if (b != b) return a; if (a == 0.0f && b == 0.0f) return std::signbit(a) ? b : a; return a > b ? a : b; }
    static pt_float4 vmin(const pt_float4& a, const pt_float4& b) { return P4(lo1(a.x, b.x), lo1(a.y, b.y), lo1(a.z, b.z)); }
    static pt_float4 vmax(const pt_float4& a, const pt_float4& b) { return P4(hi1(a.x, b.x), hi1(a.y, b.y), hi1(a.z, b.z)); }
    static float area(const pt_float4& mn, const pt_float4& mx) { float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z; return 2.0f * (dx * dy + dx * dz + dy * dz); }

    void prims() {                                               // computeInfoForBVH, main.cu:20-47
        size_t n = S.mesh.size();
        cen.resize(n); lo.resize(n); hi.resize(n);
        for (size_t i = 0; i < n; i++) {
            const pt_triangle& t = S.mesh[i];
            const pt_float4 &a = S.points[t.aInd], &b = S.points[t.bInd], &c = S.points[t.cInd];
            cen[i] = P4((a.x + b.x + c.x) / 3.0f, (a.y + b.y + c.y) / 3.0f, (a.z + b.z + c.z) / 3.0f);
            lo[i] = P4(fminf(fminf(a.x, b.x), c.x) - 0.000001f, fminf(fminf(a.y, b.y), c.y) - 0.000001f, fminf(fminf(a.z, b.z), c.z) - 0.000001f);
            hi[i] = P4(fmaxf(fmaxf(a.x, b.x), c.x) + 0.000001f, fmaxf(fmaxf(a.y, b.y), c.y) + 0.000001f, fmaxf(fmaxf(a.z, b.z), c.z) + 0.000001f);
        }
    }
    int count_left(int s, int e, int axis, float split) const { int n = 0; for (int i = s; i < e; i++) n += comp(cen[S.indices[i]], axis) < split; return n; }
    int partition(int s, int e, int axis, float split) {          // partitionPrimitives, main.cu:49-62
        int mid = s;
        for (int i = s; i < e; i++) if (comp(cen[S.indices[i]], axis) < split) { std::swap(S.indices[i], S.indices[mid]); mid++; }
        return mid;
    }
    // SAH, main.cu:64-131: 12 buckets over the node bounds; prefix / suffix sweeps give the same
    // unions and counts as the reference's nested loops, including bucket i being counted twice on
    // the right (:102-109).
    float sah_split(int s, int e, int axis, const pt_float4& mn, const pt_float4& mx) {
        constexpr int NB = 12;
        pt_float4 bmin[NB], bmax[NB]; int cnt[NB];
        for (int i = 0; i < NB; i++) { bmin[i] = P4(FLT_MAX, FLT_MAX, FLT_MAX); bmax[i] = P4(-FLT_MAX, -FLT_MAX, -FLT_MAX); cnt[i] = 0; }
        const float a0 = comp(mn, axis), ext = comp(mx, axis) - comp(mn, axis);
        for (int i = s; i < e; i++) {
            int id = S.indices[i];
            float q = NB * (comp(cen[id], axis) - a0) / ext;
            int b = (q == q && fabsf(q) < 1e9f) ? (int)q : 0;
            b = std::min(std::max(b, 0), NB - 1);
            cnt[b]++; bmin[b] = vmin(bmin[b], lo[id]); bmax[b] = vmax(bmax[b], hi[id]);
        }
        pt_float4 lmin[NB], lmax[NB], rmin[NB], rmax[NB]; int lc[NB], rc[NB];
        lmin[1] = bmin[0]; lmax[1] = bmax[0]; lc[1] = cnt[0];
        for (int i = 2; i < NB; i++) { lmin[i] = vmin(lmin[i - 1], bmin[i - 1]); lmax[i] = vmax(lmax[i - 1], bmax[i - 1]); lc[i] = lc[i - 1] + cnt[i - 1]; }
        rmin[NB - 1] = bmin[NB - 1]; rmax[NB - 1] = bmax[NB - 1]; int suffix = cnt[NB - 1]; rc[NB - 1] = cnt[NB - 1] + suffix;
        for (int i = NB - 2; i >= 1; i--) { rmin[i] = vmin(bmin[i], rmin[i + 1]); rmax[i] = vmax(bmax[i], rmax[i + 1]); suffix += cnt[i]; rc[i] = cnt[i] + suffix; }
        float best = FLT_MAX; int bestI = -1;
        const float whole = area(mn, mx);
        for (int i = 1; i < NB; i++) {
            float cost = 1.0f + (lc[i] * area(lmin[i], lmax[i]) + rc[i] * area(rmin[i], rmax[i])) / whole;
            if (cost < best && (lc[i] > 0 && rc[i] > 0)) { best = cost; bestI = i; }
        }
        if (bestI < 0) {                                          // median fallback, main.cu:119-128 (SURVEY App. D ordering)
            int mid = (s + e) / 2;
            std::sort(S.indices.begin() + s, S.indices.begin() + e, [&](int a, int b) {
                float ca = comp(cen[a], axis), cb = comp(cen[b], axis);
                return ca < cb || (!(cb < ca) && a < b);
            });
            return comp(cen[S.indices[mid]], axis);
        }
        return a0 + ext * (float(bestI) / float(NB));
    }

    void build() {
        struct Job { int s, e, parent; bool left; };
        std::vector<Job> jobs{{0, (int)S.mesh.size(), -1, false}};
        while (!jobs.empty()) {
            Job j = jobs.back(); jobs.pop_back();
            int id = (int)S.bvh.size();
            S.bvh.emplace_back();
            if (j.parent >= 0) (j.left ? S.bvh[j.parent].left : S.bvh[j.parent].right) = id;
            pt_float4 mn = lo[S.indices[j.s]], mx = hi[S.indices[j.s]];
            for (int i = j.s; i < j.e; i++) { mn = vmin(mn, lo[S.indices[i]]); mx = vmax(mx, hi[S.indices[i]]); }
            pt_bvh_node& nd = S.bvh[id];
            nd.aabbMIN = mn; nd.aabbMAX = mx;
            const int n = j.e - j.s;
            auto leaf = [&]() { nd.first = j.s; nd.primCount = n; nd.left = nd.right = -1; S.largestLeaf = std::max(S.largestLeaf, n); };
            if (n <= leafMax) { leaf(); continue; }
            float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
            int axis = (dy > dx && dy > dz) ? 1 : ((dz > dx && dz > dy) ? 2 : 0);
            float split = sah_split(j.s, j.e, axis, mn, mx);
            int nl = count_left(j.s, j.e, axis, split);
            int mid = j.s;
            if (nl > 0 && nl < n - 1) mid = partition(j.s, j.e, axis, split);
            else {                                                // centroid-mean retry, main.cu:192-202
                S.backups++;
                float sum = 0.0f;
                for (int i = j.s; i < j.e; i++) sum += comp(cen[S.indices[i]], axis);
                split = sum / n;
            }
            nl = count_left(j.s, j.e, axis, split);
            if (nl > 0 && nl < n - 1) mid = partition(j.s, j.e, axis, split);
            else { leaf(); continue; }                            // forced (possibly oversize) leaf, main.cu:213-221
            nd.primCount = 0; nd.first = -1; nd.left = nd.right = -1;
            jobs.push_back({mid, j.e, id, false});                // right is built after the whole left subtree
            jobs.push_back({j.s, mid, id, true});
        }
    }
};

int tree_depth(const std::vector<pt_bvh_node>& n) {
    int best = 0;
    std::vector<std::pair<int, int>> st{{0, 1}};
    while (!st.empty()) {
        auto [i, d] = st.back(); st.pop_back();
        best = std::max(best, d);
        if (n[i].primCount > 0) continue;
        st.push_back({n[i].left, d + 1}); st.push_back({n[i].right, d + 1});
    }
    return best;
}

// ---- camera (objects.cuh:221-264, 309-325; host libm like the reference) ------------------------
F3 rotX(F3 v, float a) { float c = cosf(a), s = sinf(a); return F3{v.x, v.y * c - v.z * s, v.y * s + v.z * c}; }
F3 rotY(F3 v, float a) { float c = cosf(a), s = sinf(a); return F3{v.x * c + v.z * s, v.y, -v.x * s + v.z * c}; }
F3 rotZ(F3 v, float a) { float c = cosf(a), s = sinf(a); return F3{v.x * c - v.y * s, v.x * s + v.y * c, v.z}; }
pt_float4 unit(F3 v) { float il = 1.0f / sqrtf(dot3(v, v)); return P4(v.x * il, v.y * il, v.z * il); }

void make_camera(bool pinhole, const float pos[3], const float rot[3], float fov, float aperture, float focal, int w, int h, pt_camera& c) {
    std::memset(&c, 0, sizeof(c));
    c.cameraOrigin = P4(pos[0], pos[1], pos[2]);
    c.w = w; c.h = h;
    c.fovScale = tanf((fov * 0.5f) * (3.141592f / 180.0f));
    c.xRot = rot[0] * (3.14159265f / 180.0f); c.yRot = rot[1] * (3.14159265f / 180.0f); c.zRot = rot[2] * (3.14159265f / 180.0f);
    c.aperture = pinhole ? 0.000001f : aperture;      // objects.cuh:234
    c.focalDist = pinhole ? 1.0f / fov : focal;       // objects.cuh:235
    c.antiAliasJitterDist = 2.0f;
    c.forward = unit(rotZ(rotY(rotX(F3{0, 0, -1}, c.xRot), c.yRot), c.zRot));
    c.right = unit(rotZ(rotY(rotX(F3{1, 0, 0}, c.xRot), c.yRot), c.zRot));
    c.up = unit(rotZ(rotY(rotX(F3{0, 1, 0}, c.xRot), c.yRot), c.zRot));
}

// ---- textures (imageUtil.cu:144-195, main.cu:364-391) ---------------------------------------------
// loadBMPToImage(path, isData = false): 24-bit BMP only; pixel rows are read straight after the two
// packed headers (54 bytes; bfOffBits is not honoured), padded to 4 bytes; row y of the file becomes
// image row height-1-y; channel/255 then powf(., 2.2f) (host libm, like the reference); alpha 1.
// Anything else — missing file included — is the reference's Image(0,0).
bool load_bmp(const std::string& file, std::vector<pt_float4>& px, int& w, int& h) {
    w = h = 0;
    FILE* f = fopen(file.c_str(), "rb");
    if (!f) return false;
    unsigned char hd[54];
    bool ok = fread(hd, 1, 54, f) == 54 && hd[0] == 'B' && hd[1] == 'M';
    uint16_t bpp = 0; int32_t bw = 0, bh = 0;
    if (ok) { std::memcpy(&bpp, hd + 28, 2); std::memcpy(&bw, hd + 18, 4); std::memcpy(&bh, hd + 22, 4); ok = bpp == 24 && bw > 0 && bh > 0; }
    // The header is untrusted input: a texture of more than 2^26 texels (1 GB as float4), or one whose pixel rows the file
    // cannot hold, is treated like a missing file (the reference's Image(0,0)) instead of overflowing `3*bw`, `bw*bh` or
    // throwing bad_alloc across the C ABI. A file that ends inside the LAST rows still loads: the reference keeps whatever
    // its row buffer held from the previous fread there, this loader zero-fills those rows (Appendix-D style definition).
    if (ok) {
        const long long texels = (long long)bw * (long long)bh, rowBytes = (3ll * bw + 3ll) & ~3ll;
        long long fileBytes = -1;
        if (fseek(f, 0, SEEK_END) == 0) fileBytes = ftell(f);
        ok = texels <= (1ll << 26) && fileBytes >= 54 + rowBytes && fseek(f, 54, SEEK_SET) == 0;
    }
    if (ok) {
        const size_t row = ((size_t)3 * (size_t)bw + 3) & ~(size_t)3;
        std::vector<unsigned char> line(row);
        px.assign((size_t)bw * bh, pt_float4{0, 0, 0, 0});
        for (int y = 0; y < bh; y++) {
            if (fread(line.data(), 1, row, f) != row) std::fill(line.begin(), line.end(), (unsigned char)0);
            for (int x = 0; x < bw; x++) {
                float b = line[x * 3 + 0] / 255.0f, g = line[x * 3 + 1] / 255.0f, r = line[x * 3 + 2] / 255.0f;
                px[(size_t)(bh - 1 - y) * bw + x] = pt_float4{powf(r, 2.2f), powf(g, 2.2f), powf(b, 2.2f), 1.0f};
            }
        }
        w = bw; h = bh;
    }
    fclose(f);
    return ok;
}

// ---- image output (imageUtil.cu:69-100, 202-232) --------------------------------------------------
float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
float aces(float c) { return clamp01((c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f)); }
// imageUtil.cu:90-92: static_cast<unsigned char>(clamp(c, 0, 1) * 255.0f + 0.5f). A NaN passes the clamp and the cast is
// undefined; x86 builds of the reference produce 0 (cvttss2si's 0x80000000, low byte) — fixed as 0 here.
unsigned char to_byte(float c) { const float v = clamp01(c) * 255.0f + 0.5f; return v != v ? (unsigned char)0 : (unsigned char)v; }

}  // namespace

extern "C" {

int pt_fail_(int code, const char* msg);             // pt_api.hip: sets pt_last_error()

static novum_scene* scene_load(const char* config_path, const char* base_dir, int render_number, int bvh_builder);

// No C++ exception may cross the C ABI (std::terminate in the host process): out-of-memory or a malformed
// file that makes a container throw ends as nullptr + pt_last_error().
novum_scene* novum_scene_load_ex(const char* config_path, const char* base_dir, int render_number, int bvh_builder) {
    try {
        return scene_load(config_path, base_dir, render_number, bvh_builder);
    } catch (const std::exception& e) {
        pt_fail_(-1, (std::string("novum_scene_load: ") + e.what()).c_str());
    } catch (...) {
        pt_fail_(-1, "novum_scene_load: unknown exception");
    }
    return nullptr;
}

static novum_scene* scene_load(const char* config_path, const char* base_dir, int render_number, int bvh_builder) {
    if (!config_path) return nullptr;
    std::unique_ptr<novum_scene> owner(new novum_scene());      // freed on every early return and on a throw
    novum_scene* S = owner.get();
    if (!parse_config(config_path, S->cfg)) { return nullptr; }
    std::string base;
    if (base_dir) base = base_dir;
    else { std::string p = config_path; size_t k = p.find_last_of('/'); base = (k == std::string::npos) ? "." : p.substr(0, k); }
    const Config& c = S->cfg;
    make_camera(c.pinhole, c.camPos, c.camRot, c.fov, c.aperture, c.focalDist, c.width, c.height, S->cam);
    // The four fixed texture files of main.cu:371-374, concatenated (main.cu:376-386); a missing
    // file is a 0x0 image (imageUtil.cu:146-149). Paths are taken relative to base_dir.
    static const char* kTextures[4] = {"textures/enkidutexture.bmp", "textures/enkiduchibitexture.bmp", "textures/leaftex2.bmp", "textures/leafautumn.bmp"};
    int tStart[4], tW[4], tH[4], cursor = 0;
    for (int i = 0; i < 4; i++) {
        std::vector<pt_float4> px;
        load_bmp(base + "/" + kTextures[i], px, tW[i], tH[i]);
        S->textures.insert(S->textures.end(), px.begin(), px.end());
        tStart[i] = cursor;
        if ((long long)cursor + (long long)tW[i] * tH[i] > 0x7fffffffll) { pt_fail_(-1, "novum_scene_load: the textures exceed 2^31 texels"); return nullptr; }
        cursor += tW[i] * tH[i];
    }
    material_table(*S, tStart, tW, tH);
    for (const MeshLine& m : c.meshes) {
        std::string p = (!m.path.empty() && m.path[0] == '/') ? m.path : base + "/" + m.path;
        float e[3] = {m.mult * m.rgb[0], m.mult * m.rgb[1], m.mult * m.rgb[2]};
        bool emissive = dot3(F3{m.rgb[0], m.rgb[1], m.rgb[2]}, F3{m.rgb[0], m.rgb[1], m.rgb[2]}) > 0.0f;
        float off[3] = {0.0f, emissive ? -0.01f * render_number : 0.0f, 0.0f};        // main.cu:476-478
        read_obj(p, *S, e, m.material, off);
    }
    if (S->mesh.empty()) { fprintf(stderr, "Error: No triangles loaded.\n"); return nullptr; }
    S->indices.resize(S->mesh.size());
    for (size_t i = 0; i < S->mesh.size(); i++) S->indices[i] = (int32_t)i;
    if (bvh_builder == NOVUM_BVH_DEVICE) {                        // f-4: same tree, built on the GPU
        S->bvh.resize(2 * S->mesh.size());
        pt_bvh_build_stats st{};
        int k = pt_bvh_build_device(S->points.data(), (int)S->points.size(), S->mesh.data(), (int)S->mesh.size(), c.leafSize,
                                    PT_BVH_REFERENCE_TREE, S->bvh.data(), (int)S->bvh.size(), S->indices.data(), &st);
        if (k <= 0) { fprintf(stderr, "novum_scene_load: %s\n", pt_last_error()); return nullptr; }
        S->bvh.resize(k);
        S->largestLeaf = st.largest_leaf; S->backups = st.backups;
    } else {
        Builder b{*S, {}, {}, {}, c.leafSize};
        b.prims();
        b.build();
    }
    S->treeDepth = tree_depth(S->bvh);
    return owner.release();
}

novum_scene* novum_scene_load(const char* config_path, const char* base_dir, int render_number) {
    return novum_scene_load_ex(config_path, base_dir, render_number, NOVUM_BVH_HOST);
}

int novum_bvh_build_host(const pt_float4* positions, int n_positions, const pt_triangle* triangles, int n_triangles,
                         int max_leaf_size, pt_bvh_node* nodes_out, int nodes_capacity, int32_t* indices_out,
                         pt_bvh_build_stats* stats) {
    if (!positions || !triangles || !nodes_out || !indices_out || n_triangles <= 0 || n_positions <= 0) return -1;
    for (int i = 0; i < n_triangles; i++) {
        const pt_triangle& t = triangles[i];
        if (t.aInd < 0 || t.aInd >= n_positions || t.bInd < 0 || t.bInd >= n_positions || t.cInd < 0 || t.cInd >= n_positions) return -1;
    }
    auto t0 = std::chrono::steady_clock::now();
    novum_scene S;
    S.points.assign(positions, positions + n_positions);
    S.mesh.assign(triangles, triangles + n_triangles);
    S.indices.resize(n_triangles);
    for (int i = 0; i < n_triangles; i++) S.indices[i] = i;
    Builder b{S, {}, {}, {}, max_leaf_size};
    b.prims();
    b.build();
    if ((int)S.bvh.size() > nodes_capacity) return -1;
    std::memcpy(nodes_out, S.bvh.data(), S.bvh.size() * sizeof(pt_bvh_node));
    std::memcpy(indices_out, S.indices.data(), sizeof(int32_t) * (size_t)n_triangles);
    if (stats) {
        *stats = pt_bvh_build_stats{};
        stats->n_nodes = (int)S.bvh.size(); stats->largest_leaf = S.largestLeaf; stats->backups = S.backups;
        stats->depth = stats->levels = tree_depth(S.bvh);
        stats->total_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return (int)S.bvh.size();
}

void novum_scene_free(novum_scene* s) { delete s; }

void novum_scene_info(const novum_scene* s, int32_t* info) {
    int v[16] = {s->cfg.width, s->cfg.height, s->cfg.spp, s->cfg.maxDepth, integrator_id(s->cfg.integrator), s->cfg.leafSize,
                 (int)s->mesh.size(), (int)s->lights.size(), (int)s->bvh.size(), (int)s->points.size(), (int)s->normals.size(),
                 (int)s->uvs.size(), (int)s->mats.size(), s->largestLeaf, s->backups, s->treeDepth};
    std::memcpy(info, v, sizeof(v));
}

void novum_scene_desc(const novum_scene* s, pt_scene_desc* d) {
    d->positions = s->points.data(); d->n_positions = (int)s->points.size();
    d->normals = s->normals.data(); d->n_normals = (int)s->normals.size();
    d->uvs = s->uvs.data(); d->n_uvs = (int)s->uvs.size();
    d->triangles = s->mesh.data(); d->n_triangles = (int)s->mesh.size();
    d->lights = s->lights.data(); d->n_lights = (int)s->lights.size();
    d->bvh = s->bvh.data(); d->n_nodes = (int)s->bvh.size();
    d->bvh_indices = s->indices.data();
    d->materials = s->mats.data(); d->n_materials = (int)s->mats.size();
    d->textures = s->textures.data(); d->n_texels = (int)s->textures.size();
}

void novum_scene_camera(const novum_scene* s, pt_camera* out) { *out = s->cam; }

void novum_make_camera(int pinhole, const float* pos, const float* rot, float fov, float aperture, float focal, int w, int h, pt_camera* out) {
    make_camera(pinhole != 0, pos, rot, fov, aperture, focal, w, h, *out);
}

void novum_finalise(float* rgba, int n, int spp) {                 // main.cu:860-870
    const float s = (float)spp;
    for (int i = 0; i < n; i++) {
        float* p = rgba + 4 * (size_t)i;
        p[0] /= s; p[1] /= s; p[2] /= s;
        if (std::isnan(p[0]) || std::isnan(p[1]) || std::isnan(p[2])) { p[0] = 1.0f; p[1] = 0.0f; p[2] = 1.0f; p[3] = 0.0f; }
        if (std::isinf(p[0]) || std::isinf(p[1]) || std::isinf(p[2])) { p[0] = 0.0f; p[1] = 1.0f; p[2] = 0.0f; p[3] = 0.0f; }
    }
}

int novum_save_bmp(const char* path, const float* rgba, int w, int h, int post) {      // imageUtil.cu:69-100, 234-257
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    const int row = (3 * w + 3) & ~3;
    const uint32_t img = (uint32_t)row * h, off = 54, size = off + img;
    unsigned char hd[54] = {0};
    hd[0] = 'B'; hd[1] = 'M';
    std::memcpy(hd + 2, &size, 4); std::memcpy(hd + 10, &off, 4);
    uint32_t ih = 40; std::memcpy(hd + 14, &ih, 4);
    int32_t ww = w, hh = h; std::memcpy(hd + 18, &ww, 4); std::memcpy(hd + 22, &hh, 4);
    uint16_t planes = 1, bpp = 24; std::memcpy(hd + 26, &planes, 2); std::memcpy(hd + 28, &bpp, 2);
    std::memcpy(hd + 34, &img, 4);
    fwrite(hd, 1, 54, f);
    std::vector<unsigned char> line(row, 0);
    const float ig = 1.0f / 2.2f;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const float* p = rgba + 4 * ((size_t)y * w + x);
            float r = p[0], g = p[1], b = p[2];
            if (post) { r = powf(aces(r), ig); g = powf(aces(g), ig); b = powf(aces(b), ig); }      // toneMap + gammaCorrect
            line[x * 3 + 0] = to_byte(b);
            line[x * 3 + 1] = to_byte(g);
            line[x * 3 + 2] = to_byte(r);
        }
        fwrite(line.data(), 1, row, f);
    }
    fclose(f);
    return 0;
}

int novum_save_csv_mono(const char* path, const float* rgba, int w, int h, int channel) {      // imageUtil.cu:123-142
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            fprintf(f, "%.3e", (double)rgba[4 * ((size_t)y * w + x) + channel]);       // std::scientific, precision 3
            if (x < w - 1) fputc(',', f);
        }
        fputc('\n', f);
    }
    fclose(f);
    return 0;
}

// hipMalloc / hipMemset / hipMemcpy / hipFree without pulling the HIP headers into this file
int pt_host_alloc_zero_(void** p, size_t bytes);
int pt_host_download_(void* d, void* h, size_t bytes);
int pt_host_download_free_(void* d, void* h, size_t bytes);

namespace {
struct Preview {                       // the `elapsed >= saveIntervalSeconds` block of deviceCode.cu:574-604
    void* dColors; std::vector<float>* host; int w, h; bool post;
    const char* bmp; const char* csv; double interval; double last;
};
double now_seconds() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int preview_cb(int samplesDone, void* user) {
    Preview* pv = (Preview*)user;
    double t = now_seconds();
    if (t - pv->last < pv->interval) return 0;
    if (pt_host_download_(pv->dColors, pv->host->data(), (size_t)pv->w * pv->h * 16) != 0) return 0;
    novum_finalise(pv->host->data(), pv->w * pv->h, samplesDone);      // h_colors / (currSample + 1), NaN / Inf painted
    if (pv->bmp) novum_save_bmp(pv->bmp, pv->host->data(), pv->w, pv->h, pv->post ? 1 : 0);
    if (pv->csv) novum_save_csv_mono(pv->csv, pv->host->data(), pv->w, pv->h, 0);
    pv->last = t;
    return 0;
}
}  // namespace

int novum_init_render_progressive(const char* config_path, const char* base_dir, int render_number, float* out_rgba, const char* bmp_path,
                                  const char* preview_bmp, const char* preview_csv, double interval_seconds, int chunk_spp) {
    novum_scene* S = novum_scene_load(config_path, base_dir, render_number);
    if (!S) return -1;
    int integ = integrator_id(S->cfg.integrator);
    if (integ != PT_UNIDIRECTIONAL && integ != PT_NAIVE_UNIDIRECTIONAL) {
        fprintf(stderr, "novum_init_render: integrator '%s' is outside the accelerated path\n", S->cfg.integrator.c_str());
        novum_scene_free(S);
        return -3;
    }
    pt_scene_desc d;
    novum_scene_desc(S, &d);
    pt_scene* dev = pt_scene_create(&d);
    if (!dev) { fprintf(stderr, "novum_init_render: %s\n", pt_last_error()); novum_scene_free(S); return -2; }
    const int w = S->cfg.width, h = S->cfg.height;
    const size_t bytes = (size_t)w * h * 16;
    void* colors = nullptr;                                            // out_colors, main.cu:337-339
    int rc = pt_host_alloc_zero_(&colors, bytes);
    std::vector<float> host((size_t)w * h * 4);
    if (rc == 0) {
        if (chunk_spp > 0 && (preview_bmp || preview_csv)) {
            Preview pv{colors, &host, w, h, S->cfg.postProcess, preview_bmp, preview_csv, interval_seconds, now_seconds()};
            rc = pt_launch_progressive(integ, S->cfg.maxDepth, S->cam, dev, S->cfg.spp, 1, w, h, colors, chunk_spp, preview_cb, &pv);
        } else {
            rc = (integ == PT_UNIDIRECTIONAL) ? pt_launch_unidirectional(S->cfg.maxDepth, S->cam, dev, S->cfg.spp, 1, w, h, colors)     // main.cu:565
                                              : pt_launch_naive_unidirectional(S->cfg.maxDepth, S->cam, dev, S->cfg.spp, 1, w, h, colors);   // main.cu:677
        }
        int rc2 = pt_host_download_free_(colors, host.data(), bytes);  // main.cu:854-855, 889
        if (rc == 0) rc = rc2;
    }
    if (rc == 0) {
        novum_finalise(host.data(), w * h, S->cfg.spp);
        if (out_rgba) std::memcpy(out_rgba, host.data(), bytes);
        if (bmp_path) rc = novum_save_bmp(bmp_path, host.data(), w, h, S->cfg.postProcess ? 1 : 0);
    } else fprintf(stderr, "novum_init_render: %s\n", pt_last_error());
    pt_scene_destroy(dev);
    novum_scene_free(S);
    return rc;
}

int novum_init_render(const char* config_path, const char* base_dir, int render_number, float* out_rgba, const char* bmp_path) {
    return novum_init_render_progressive(config_path, base_dir, render_number, out_rgba, bmp_path, nullptr, nullptr, 0.0, 0);
}

}  // extern "C"
