// oracle_capi.cpp — TEST INFRASTRUCTURE (see oracle/README.md). PARITY UNPINNED.
// Plain-C entry points, loaded with ctypes by tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg only.
#include <chrono>
#include <cstring>

#include "oracle.h"

using namespace oracle;

struct OracleScene {
    RenderConfig cfg;
    Scene sc;
    Camera cam;
};

extern "C" {

void* oracle_scene_load(const char* configPath, const char* baseDir, int renderNumber) {
    OracleScene* s = new OracleScene();
    if (!loadSceneFromConfig(configPath, baseDir ? baseDir : "", renderNumber, s->cfg, s->sc, s->cam)) {
        delete s;
        return nullptr;
    }
    return s;
}

// The scene straight from arrays in the reference's data model (what initRender uploads,
// main.cu:469-557): lets tests hand-build trees the SAH builder would never produce.
void* oracle_scene_from_arrays(const void* points, int nPoints, const void* normals, int nNormals, const void* uvs, int nUvs,
                               const void* mesh, int nTris, const void* lights, int nLights, const void* bvh, int nNodes,
                               const int* indices, const void* mats, int nMats, const void* textures, int nTexels) {
    OracleScene* s = new OracleScene();
    s->sc.points.assign((const float4*)points, (const float4*)points + nPoints);
    s->sc.normals.assign((const float4*)normals, (const float4*)normals + nNormals);
    s->sc.uvs.assign((const float2*)uvs, (const float2*)uvs + nUvs);
    s->sc.mesh.assign((const Triangle*)mesh, (const Triangle*)mesh + nTris);
    s->sc.lights.assign((const Triangle*)lights, (const Triangle*)lights + nLights);
    s->sc.bvh.assign((const BVHnode*)bvh, (const BVHnode*)bvh + nNodes);
    s->sc.indices.assign(indices, indices + nTris);
    s->sc.mats.assign((const Material*)mats, (const Material*)mats + nMats);
    if (textures && nTexels > 0) s->sc.textures.assign((const float4*)textures, (const float4*)textures + nTexels);
    s->cfg.integratorType = "UNIDIRECTIONAL";
    s->cam = cameraPinhole(f4(0.0f, 0.0f, 1.0f), 8, 8, 0.0f, 0.0f, 0.0f, 60.0f);
    return s;
}

// The reference's builder alone (computeInfoForBVH + buildBVH, main.cu:20-233, call site :524-530)
// on caller arrays: nodesOut has room for 2*nTris-1 BVHnodes, indicesOut for nTris ints.
// stats[0..3] = node count, largest leaf, backup count, tree depth. Returns the node count.
int oracle_build_bvh(const void* points, int nPoints, const void* mesh, int nTris, int maxLeafSize,
                     void* nodesOut, int* indicesOut, int* stats) {
    Scene sc;
    sc.points.assign((const float4*)points, (const float4*)points + nPoints);
    sc.mesh.assign((const Triangle*)mesh, (const Triangle*)mesh + nTris);
    buildSceneBVH(sc, maxLeafSize);
    std::memcpy(nodesOut, sc.bvh.data(), sc.bvh.size() * sizeof(BVHnode));
    std::memcpy(indicesOut, sc.indices.data(), sc.indices.size() * sizeof(int));
    if (stats) { stats[0] = (int)sc.bvh.size(); stats[1] = sc.largestLeaf; stats[2] = sc.backupCount; stats[3] = sc.maxDepthOfTree; }
    return (int)sc.bvh.size();
}

void oracle_scene_free(void* h) { delete (OracleScene*)h; }

int oracle_scene_texels(void* h) { return (int)((OracleScene*)h)->sc.textures.size(); }

// info[0..15]: width,height,spp,maxDepth,integrator,leafSize,nTris,nLights,nNodes,nPoints,
//              nNormals,nUvs,nMats,largestLeaf,backupCount,treeDepth
void oracle_scene_info(void* h, int* info) {
    OracleScene* s = (OracleScene*)h;
    int integ = -1;
    const std::string& n = s->cfg.integratorType;       // objects.cuh:583-593
    if (n == "UNIDIRECTIONAL") integ = 0; else if (n == "BIDIRECTIONAL" || n == "BDPT") integ = 1;
    else if (n == "NAIVE_UNIDIRECTIONAL") integ = 2; else if (n == "VCM") integ = 3; else if (n == "SPPM") integ = 4;
    int v[16] = {s->cfg.width, s->cfg.height, s->cfg.sampleCount, s->cfg.maxDepth, integ, s->cfg.bvhLeafSize,
                 (int)s->sc.mesh.size(), (int)s->sc.lights.size(), (int)s->sc.bvh.size(), (int)s->sc.points.size(),
                 (int)s->sc.normals.size(), (int)s->sc.uvs.size(), (int)s->sc.mats.size(), s->sc.largestLeaf,
                 s->sc.backupCount, s->sc.maxDepthOfTree};
    std::memcpy(info, v, sizeof(v));
}

// what: 0 points(float4) 1 normals(float4) 2 uvs(float2) 3 mesh(Triangle 80B) 4 lights(Triangle)
//       5 bvh(BVHnode 48B) 6 indices(int) 7 materials(176B) 8 camera(112B)
void oracle_scene_get(void* h, int what, void* dst) {
    OracleScene* s = (OracleScene*)h;
    switch (what) {
        case 0: std::memcpy(dst, s->sc.points.data(), s->sc.points.size() * sizeof(float4)); break;
        case 1: std::memcpy(dst, s->sc.normals.data(), s->sc.normals.size() * sizeof(float4)); break;
        case 2: std::memcpy(dst, s->sc.uvs.data(), s->sc.uvs.size() * sizeof(float2)); break;
        case 3: std::memcpy(dst, s->sc.mesh.data(), s->sc.mesh.size() * sizeof(Triangle)); break;
        case 4: std::memcpy(dst, s->sc.lights.data(), s->sc.lights.size() * sizeof(Triangle)); break;
        case 5: std::memcpy(dst, s->sc.bvh.data(), s->sc.bvh.size() * sizeof(BVHnode)); break;
        case 6: std::memcpy(dst, s->sc.indices.data(), s->sc.indices.size() * sizeof(int)); break;
        case 7: std::memcpy(dst, s->sc.mats.data(), s->sc.mats.size() * sizeof(Material)); break;
        case 8: std::memcpy(dst, &s->cam, sizeof(Camera)); break;
        case 9: std::memcpy(dst, s->sc.textures.data(), s->sc.textures.size() * sizeof(float4)); break;
        default: break;
    }
}

void oracle_make_camera(int pinhole, const float* pos, const float* rot, float fov, float aperture, float focalDist, int w, int h, void* out) {
    Camera c = pinhole ? cameraPinhole(f4(pos[0], pos[1], pos[2]), w, h, rot[0], rot[1], rot[2], fov)
                       : cameraNotPinhole(f4(pos[0], pos[1], pos[2]), w, h, rot[0], rot[1], rot[2], fov, aperture, focalDist);
    std::memcpy(out, &c, sizeof(c));
}

// colors: w*h float4 accumulator (+= semantics); counters: w*h x 8 uint32 or NULL.
// Returns wall seconds spent in the render loop.
double oracle_render(void* h, const void* cam, int w, int hgt, int spp, int maxDepth, int integrator, int useMIS,
                     unsigned long long seed, int x0, int y0, int x1, int y1, float* colors, unsigned int* counters, int nThreads) {
    OracleScene* s = (OracleScene*)h;
    Camera c; std::memcpy(&c, cam, sizeof(c));
    auto t0 = std::chrono::steady_clock::now();
    launch(integrator, maxDepth, c, s->sc, spp, useMIS != 0, w, hgt, seed, x0, y0, x1, y1, (float4*)colors, (PixelCounters*)counters, nThreads);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void oracle_finalise(float* colors, int n, int spp) { finalise((float4*)colors, n, spp); }

// ---- RNG ----
void oracle_xorwow_init(unsigned long long seed, unsigned int subsequence, unsigned int* state6) {
    XorwowState st; xorwow_init(st, seed, subsequence);
    std::memcpy(state6, st.v, 20); state6[5] = st.d;
}
void oracle_xorwow_next(unsigned int* state6, int n, unsigned int* out) {
    XorwowState st; std::memcpy(st.v, state6, 20); st.d = state6[5];
    for (int i = 0; i < n; i++) out[i] = xorwow_next(st);
    std::memcpy(state6, st.v, 20); state6[5] = st.d;
}
void oracle_xorwow_uniform(unsigned int* state6, int n, float* out) {
    XorwowState st; std::memcpy(st.v, state6, 20); st.d = state6[5];
    for (int i = 0; i < n; i++) out[i] = xorwow_uniform(st);
    std::memcpy(state6, st.v, 20); state6[5] = st.d;
}
// k = -1: one-step matrix A; k >= 0: A^(2^67 * 2^k). out: 800 words.
void oracle_xorwow_matrix(int k, unsigned int* out) {
    if (k < 0) { XorwowMatrix a; xorwow_step_matrix(a); std::memcpy(out, a.row, sizeof(a.row)); }
    else std::memcpy(out, xorwow_jump_table().jump[k].row, sizeof(XorwowMatrix));
}

// ---- math ----
void oracle_sincosf(int n, const float* x, float* s, float* c) { for (int i = 0; i < n; i++) ref_sincosf(x[i], &s[i], &c[i]); }
void oracle_expf(int n, const float* x, float* y) { for (int i = 0; i < n; i++) y[i] = ref_expf(x[i]); }
void oracle_rsqrtf(int n, const float* x, float* y) { for (int i = 0; i < n; i++) y[i] = ref_rsqrtf(x[i]); }
void oracle_pow5(int n, const float* x, float* y) { for (int i = 0; i < n; i++) y[i] = ref_pow5(x[i]); }

// ---- traversal probes: rays = n x 6 floats (origin, direction) ----
// out_i: n x 4 (valid, triIDX, materialID, backface); out_f: n x 12 (t,u,v, point.xyz, normal.xyz, uv.xy, 0)
void oracle_trace_closest(void* h, int n, const float* rays, int bruteForce, int* out_i, float* out_f, unsigned int* counters8) {
    OracleScene* s = (OracleScene*)h;
    PixelCounters pc; std::memset(&pc, 0, sizeof(pc));
    for (int i = 0; i < n; i++) {
        Ray r; r.origin = f4(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]); r.direction = f4(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        Intersection it;
        if (bruteForce) sceneIntersection(r, s->sc, it); else BVHSceneIntersect(r, s->sc, it, 999999.0f, -1, &pc);
        out_i[4 * i] = it.valid; out_i[4 * i + 1] = it.valid ? it.triIDX : -1; out_i[4 * i + 2] = it.valid ? it.materialID : -1; out_i[4 * i + 3] = it.valid ? it.backface : 0;
        float* o = out_f + 12 * i;
        std::memset(o, 0, 12 * sizeof(float));
        if (it.valid) {
            o[0] = it.dist; o[1] = it.baryU; o[2] = it.baryV; o[3] = it.point.x; o[4] = it.point.y; o[5] = it.point.z;
            o[6] = it.normal.x; o[7] = it.normal.y; o[8] = it.normal.z; o[9] = it.uv.x; o[10] = it.uv.y;
        }
    }
    if (counters8) std::memcpy(counters8, &pc, sizeof(pc));
}
// out_f: n x 3 throughputScale
void oracle_trace_shadow(void* h, int n, const float* rays, const float* max_t, float* out_f, unsigned int* counters8) {
    OracleScene* s = (OracleScene*)h;
    PixelCounters pc; std::memset(&pc, 0, sizeof(pc));
    for (int i = 0; i < n; i++) {
        Ray r; r.origin = f4(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]); r.direction = f4(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        float4 ts;
        BVHShadowRay(r, s->sc, ts, max_t[i], -1, &pc);
        out_f[3 * i] = ts.x; out_f[3 * i + 1] = ts.y; out_f[3 * i + 2] = ts.z;
    }
    if (counters8) std::memcpy(counters8, &pc, sizeof(pc));
}

// ---- camera rays: for pixel (x,y), stream (seed, y*w+x): out 6 floats ----
void oracle_camera_ray(const void* cam, unsigned long long seed, int x, int y, float* out6) {
    Camera c; std::memcpy(&c, cam, sizeof(c));
    XorwowState st; xorwow_init(st, seed, (uint32_t)(y * c.w + x));
    Ray r = generateCameraRay(c, st, x, y, nullptr);
    out6[0] = r.origin.x; out6[1] = r.origin.y; out6[2] = r.origin.z; out6[3] = r.direction.x; out6[4] = r.direction.y; out6[5] = r.direction.z;
}

// ---- BSDF probes. wi is the local direction INTO the surface (as the integrator passes it). ----
// sample: out 8 floats (wo.xyz, f.xyz, pdf, nDraws)
void oracle_bsdf_sample(void* h, int materialID, const float* wi3, int backface, float etaI, float etaT, unsigned long long seed, unsigned int subseq, float* out8) {
    OracleScene* s = (OracleScene*)h;
    XorwowState st; xorwow_init(st, seed, subseq);
    PixelCounters pc; std::memset(&pc, 0, sizeof(pc));
    float4 wo = f4(), f = f4(); float pdf = 0.0f;
    sample_f_eval(st, s->sc, materialID, f4(wi3[0], wi3[1], wi3[2]), etaI, etaT, backface != 0, wo, f, pdf, f2(0.0f, 0.0f), &pc);
    out8[0] = wo.x; out8[1] = wo.y; out8[2] = wo.z; out8[3] = f.x; out8[4] = f.y; out8[5] = f.z; out8[6] = pdf; out8[7] = (float)pc.rngDraws;
}
// eval: out 4 floats (f.xyz, pdf)
void oracle_bsdf_eval(void* h, int materialID, const float* wi3, const float* wo3, float etaI, float etaT, float* out4) {
    OracleScene* s = (OracleScene*)h;
    float4 f = f4(); float pdf = 0.0f;
    f_eval(s->sc, materialID, f4(wi3[0], wi3[1], wi3[2]), f4(wo3[0], wo3[1], wo3[2]), etaI, etaT, f, f2(0.0f, 0.0f));
    pdf_eval(s->sc, materialID, f4(wi3[0], wi3[1], wi3[2]), f4(wo3[0], wo3[1], wo3[2]), etaI, etaT, pdf, f2(0.0f, 0.0f));
    out4[0] = f.x; out4[1] = f.y; out4[2] = f.z; out4[3] = pdf;
}

}  // extern "C"
