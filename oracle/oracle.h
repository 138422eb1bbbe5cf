// oracle.h — TEST INFRASTRUCTURE (see oracle/README.md). PARITY UNPINNED.
// Internal declarations shared by the oracle's translation units.
#pragma once
#include "oracle_math.h"
#include "oracle_rng.h"
#include "oracle_types.h"

namespace oracle {

// ---- oracle_scene.cpp -------------------------------------------------------
bool loadConfig(const std::string& filepath, RenderConfig& config);                 // objects.cuh:844-943
void readObjSimple(const std::string& filename, Scene& sc, float4 e, int materialID, float4 offset);  // main.cu:936-1068
void buildMaterialTable(Scene& sc, const int startIndices[4], const int widths[4], const int heights[4]);   // main.cu:397-467
void loadTextures(Scene& sc, const std::string& baseDir, int startIndices[4], int widths[4], int heights[4]);   // main.cu:364-391 + imageUtil.cu:144-195
void buildSceneBVH(Scene& sc, int maxLeafSize);                                     // main.cu:20-233, 502-530
Camera cameraPinhole(const float4& origin, int w, int h, float xR, float yR, float zR, float FOV, float aajitter = 2.0f);   // objects.cuh:221-242
Camera cameraNotPinhole(const float4& origin, int w, int h, float xR, float yR, float zR, float FOV, float aperture, float focalDist, float aajitter = 2.0f);  // objects.cuh:244-264
// initRender up to (not including) the launcher call, main.cu:235-557; returns false on error
bool loadSceneFromConfig(const std::string& configPath, const std::string& baseDir, int renderNumber,
                         RenderConfig& cfg, Scene& sc, Camera& cam);

// ---- oracle_render.cpp ------------------------------------------------------
bool triangleIntersect(const Vertices& verts, const Triangle& tri, const Ray& r, float4& barycentric, float& tval);  // integratorUtilities.cuh:8-42
bool aabbIntersect(const Ray& r, float4 minCorner, float4 maxCorner, float& tmin, float& tmax);                     // integratorUtilities.cuh:44-82
void BVHSceneIntersect(const Ray& r, const Scene& sc, Intersection& intersect, float max_t, int skipTri, PixelCounters* pc);  // integratorUtilities.cuh:84-186
void BVHShadowRay(const Ray& r, const Scene& sc, float4& throughputScale, float max_t, int skip_tri, PixelCounters* pc);      // integratorUtilities.cuh:188-288
void sceneIntersection(const Ray& r, const Scene& sc, Intersection& intersect);                                     // integratorUtilities.cuh:290-335
Ray generateCameraRay(const Camera& cam, XorwowState& st, int x, int y, PixelCounters* pc);                        // objects.cuh:268-307

void f_eval(const Scene& sc, int materialID, const float4& wi, const float4& wo, float etaI, float etaT, float4& f_val, float2 uv, int transportMode = TRANSPORTMODE_RADIANCE);   // reflectors.cuh:547-584
void sample_f_eval(XorwowState& st, const Scene& sc, int materialID, const float4& wi, float etaI, float etaT, bool backface, float4& wo, float4& f_val, float& pdf, float2 uv, PixelCounters* pc, int transportMode = TRANSPORTMODE_RADIANCE);  // reflectors.cuh:588-629
void pdf_eval(const Scene& sc, int materialID, const float4& wi, const float4& wo, float etaI, float etaT, float& pdf, float2 uv);   // reflectors.cuh:633-666

// One sample of one pixel (the body of the reference kernels, deviceCode.cu:158-205 / 285-542).
float4 Li_naive_unidirectional(XorwowState& st, const Camera& cam, const Scene& sc, int maxDepth, int x, int y, PixelCounters* pc);
float4 Li_unidirectional(XorwowState& st, const Camera& cam, const Scene& sc, int maxDepth, bool useMIS, int x, int y, PixelCounters* pc);

// launch_unidirectional / launch_naive_unidirectional (deviceCode.cu:544-620 / 207-283) over the
// pixel rectangle [x0,x1) x [y0,y1) of a w x h image. colors is the w*h accumulator (sum over
// samples, += semantics); counters (optional) is w*h. nThreads host threads split the rows.
void launch(int integrator, int maxDepth, const Camera& cam, const Scene& sc, int numSample, bool useMIS,
            int w, int h, uint64_t seed, int x0, int y0, int x1, int y1,
            float4* colors, PixelCounters* counters, int nThreads);

// main.cu:860-870 finalise: divide by spp, NaN -> (1,0,1), Inf -> (0,1,0)
void finalise(float4* colors, int n, int sampleCount);

}  // namespace oracle
