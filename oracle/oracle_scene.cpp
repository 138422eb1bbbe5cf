// oracle_scene.cpp — TEST INFRASTRUCTURE (see oracle/README.md). PARITY UNPINNED.
//
// Restatement of the reference's scene loader: `.rendertron` config parser, OBJ reader,
// hard-coded material table, SAH BVH builder and camera set-up — everything `initRender`
// (main.cu:235-557) does before it calls the launcher. Follows the reference's control flow
// line by line so that the product's loader (cudapathtracer_amd/csrc/host) can be checked
// against it array for array.
#include <algorithm>
#include <cfloat>
#include <fstream>
#include <iostream>
#include <sstream>

#include "oracle.h"

namespace oracle {

// util.cuh:288-306
static std::string trim(const std::string& str) {
    size_t first = str.find_first_not_of(" \t\r\n");
    if (std::string::npos == first) return str;
    size_t last = str.find_last_not_of(" \t\r\n");
    return str.substr(first, (last - first + 1));
}
static float4 parseVec3(const std::string& val) {
    float4 v = f4();                       // reference leaves .w uninitialised (util.cuh:296); defined 0 here
    std::stringstream ss(val);
    ss >> v.x >> v.y >> v.z;
    return v;
}
static bool parseBool(const std::string& val) {
    std::string v = val;
    std::transform(v.begin(), v.end(), v.begin(), ::tolower);
    return v == "true";
}

// objects.cuh:844-943. Unknown keys are ignored; everything after a line starting with
// "Meshes" is a mesh line `path; mult * (r, g, b); materialID`.
bool loadConfig(const std::string& filepath, RenderConfig& config) {
    std::ifstream file(filepath);
    if (!file.is_open()) {
        std::cerr << "Error: Could not open config file: " << filepath << std::endl;
        return false;
    }
    std::string line;
    bool parsingMeshes = false;
    while (std::getline(file, line)) {
        line = trim(line);
        if (line.empty()) continue;
        if (line.rfind("Meshes", 0) == 0) { parsingMeshes = true; continue; }
        if (parsingMeshes) {
            MeshConfig mesh;
            std::stringstream ss(line);
            std::string segment;
            if (std::getline(ss, segment, ';')) mesh.path = trim(segment);
            if (std::getline(ss, segment, ';')) {
                std::string complexEm = trim(segment);
                size_t starPos = complexEm.find('*');
                size_t openParen = complexEm.find('(');
                size_t closeParen = complexEm.find(')');
                if (starPos != std::string::npos && openParen != std::string::npos) {
                    mesh.emissionMultiplier = std::stof(complexEm.substr(0, starPos));
                    std::string vecStr = complexEm.substr(openParen + 1, closeParen - openParen - 1);
                    std::replace(vecStr.begin(), vecStr.end(), ',', ' ');
                    mesh.emissionColor = parseVec3(vecStr);
                }
            }
            if (std::getline(ss, segment, ';')) mesh.materialID = std::stoi(trim(segment));
            config.meshes.push_back(mesh);
        } else {
            size_t delimiterPos = line.find(':');
            if (delimiterPos == std::string::npos) continue;
            std::string key = trim(line.substr(0, delimiterPos));
            std::string value = trim(line.substr(delimiterPos + 1));
            if (value.empty()) continue;
            if (key == "width") config.width = std::stoi(value);
            else if (key == "height") config.height = std::stoi(value);
            else if (key == "Integrator") config.integratorType = value;
            else if (key == "Name") config.name = value;
            else if (key == "Sample Count") config.sampleCount = std::stoi(value);
            else if (key == "Unidirectional Max Depth") config.maxDepth = std::stoi(value);
            else if (key == "BVH recommended leaf size") config.bvhLeafSize = std::stoi(value);
            else if (key == "Pinhole Camera") config.pinholeCamera = parseBool(value);
            else if (key == "Post Process") config.postProcess = parseBool(value);
            else if (key == "Camera Position") config.camPos = parseVec3(value);
            else if (key == "Camera Rotation") config.camRot = parseVec3(value);
            else if (key == "Camera FOV") config.camFov = std::stof(value);
            else if (key == "Camera Apeture") config.camApeture = std::stof(value);
            else if (key == "Camera FocalDist") config.camFocalDist = std::stof(value);
            // BDPT / VCM keys (objects.cuh:916-939) are out of scope and ignored like unknown keys
        }
    }
    return true;
}

// util.cuh:128-131 with rsqrtf := 1/sqrt (oracle_math.h)
static inline float4 normalize(const float4& v) {
    float invLen = ref_rsqrtf(dot(v, v));
    return f4(v.x * invLen, v.y * invLen, v.z * invLen, 0.0f);
}

// main.cu:936-1068. Fan triangulation, degenerate-triangle cull (areaSq < 1e-18), emissive
// meshes also fill the light list. SURVEY App. D: a face without `vn` gets its geometric
// normal synthesised (reference would index normals[-1]); without `vt` it gets uv (0,0).
void readObjSimple(const std::string& filename, Scene& sc, float4 e, int materialID, float4 offset) {
    std::ifstream file(filename);
    if (!file.is_open()) {
        std::cerr << "Error: Could not open OBJ file\n";
        return;
    }
    std::vector<float4>& points = sc.points;
    std::vector<float4>& normals = sc.normals;
    std::vector<float2>& uvs = sc.uvs;
    std::vector<Triangle>& mesh = sc.mesh;
    std::vector<Triangle>& lights = sc.lights;

    int startIndex = (int)points.size();
    int normalStartIndex = (int)normals.size();
    int uvStartIndex = (int)uvs.size();
    int nextLightIndex = (int)lights.size();
    int zeroUvIndex = -1;

    std::string line;
    while (std::getline(file, line)) {
        if (line.empty() || line[0] == '#' || line[0] == 's') continue;
        std::istringstream iss(line);
        std::string prefix;
        iss >> prefix;
        if (prefix == "v") {
            double x, y, z;
            iss >> x >> y >> z;
            float4 p = f4((float)x, (float)y, (float)z, 0.0f) + offset;
            points.push_back(p);
        } else if (prefix == "vt") {
            double u, v;
            iss >> u >> v;
            uvs.push_back(f2((float)u, (float)(1.0 - v)));     // `1.0f - v` with v double: double subtract
        } else if (prefix == "vn") {
            double x, y, z;
            iss >> x >> y >> z;
            if (iss.fail() || std::isnan(x) || std::isnan(y) || std::isnan(z)) {
                normals.push_back(f4(0.0f, 1.0f, 0.0f, 0.0f));
                continue;
            }
            float4 n = f4((float)x, (float)y, (float)z, 0.0f);
            float lenSq = n.x * n.x + n.y * n.y + n.z * n.z;
            if (lenSq < 1e-12f) n = f4(0.0f, 1.0f, 0.0f, 0.0f);
            normals.push_back(n);
        } else if (prefix == "f") {
            std::string vertinfo;
            std::vector<int> vertexIndices, normalIndices, uvIndices;
            while (iss >> vertinfo) {
                std::istringstream vss(vertinfo);
                std::string idx;
                if (std::getline(vss, idx, '/')) { if (!idx.empty()) vertexIndices.push_back(std::stoi(idx) - 1); }
                if (std::getline(vss, idx, '/')) { if (!idx.empty()) uvIndices.push_back(std::stoi(idx) - 1); }
                if (std::getline(vss, idx, '/')) { if (!idx.empty()) normalIndices.push_back(std::stoi(idx) - 1); }
            }
            bool hasUV = uvIndices.size() == vertexIndices.size();
            bool hasN = normalIndices.size() == vertexIndices.size();
            int n = (int)vertexIndices.size();
            for (int i = 1; i < n - 1; ++i) {
                bool isLight = lengthSquared(e) > 0;
                int idx0 = vertexIndices[0] + startIndex;
                int idx1 = vertexIndices[i] + startIndex;
                int idx2 = vertexIndices[i + 1] + startIndex;
                float4 p0 = points[idx0], p1 = points[idx1], p2 = points[idx2];
                float4 e1 = f4(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z);
                float4 e2 = f4(p2.x - p0.x, p2.y - p0.y, p2.z - p0.z);
                float4 cp = cross3(e1, e2);
                float areaSq = dot(cp, cp);
                if (areaSq < 1e-18f) continue;

                int uv_idx0, uv_idx1, uv_idx2, n_idx0, n_idx1, n_idx2;
                if (hasUV) {
                    uv_idx0 = uvIndices[0] + uvStartIndex; uv_idx1 = uvIndices[i] + uvStartIndex; uv_idx2 = uvIndices[i + 1] + uvStartIndex;
                } else {
                    if (zeroUvIndex < 0) { zeroUvIndex = (int)uvs.size(); uvs.push_back(f2(0.0f, 0.0f)); }
                    uv_idx0 = uv_idx1 = uv_idx2 = zeroUvIndex;
                }
                if (hasN) {
                    n_idx0 = normalIndices[0] + normalStartIndex; n_idx1 = normalIndices[i] + normalStartIndex; n_idx2 = normalIndices[i + 1] + normalStartIndex;
                } else {
                    n_idx0 = n_idx1 = n_idx2 = (int)normals.size();
                    normals.push_back(normalize(cp));
                }
                Triangle tri;
                tri.aInd = idx0; tri.bInd = idx1; tri.cInd = idx2;
                tri.naInd = n_idx0; tri.nbInd = n_idx1; tri.ncInd = n_idx2;
                tri.uvaInd = uv_idx0; tri.uvbInd = uv_idx1; tri.uvcInd = uv_idx2;
                tri.materialID = materialID;
                tri.emission = e;
                tri.lightInd = isLight ? nextLightIndex : -51;
                tri.triInd = (int)mesh.size();
                mesh.push_back(tri);
                if (isLight) { lights.push_back(tri); nextLightIndex++; }
            }
        }
    }
}

// ---- material factories, objects.cuh:640-791. SURVEY App. D: every Material is zero-
// initialised before the constructor defaults and the factory's assignments apply. --------
static Material matDefault() {
    Material m;
    std::memset(&m, 0, sizeof(m));
    m.type = MAT_DIFFUSE; m.albedo = f4(0.8f); m.roughness = 0.5f; m.eta = f4(0); m.k = f4(0);
    m.ior = 1.5f; m.metallic = 0.0f; m.specular = 1.0f; m.transmission = 0.0f;
    return m;
}
static Material matDiffuse(const float4& color) {
    Material m = matDefault();
    m.type = MAT_DIFFUSE; m.hasTexture = false; m.albedo = color; m.roughness = 1.0f;
    m.boundary = false; m.absorption = f4(); m.thinWalled = false; m.isSpecular = false;
    return m;
}
static Material matDiffuseTextured(int sInd, int w, int h) {
    Material m = matDefault();
    m.type = MAT_DIFFUSE; m.hasTexture = true; m.startInd = sInd; m.width = w; m.height = h;
    m.roughness = 1.0f; m.boundary = false; m.absorption = f4(); m.thinWalled = false; m.isSpecular = false;
    return m;
}
static Material matMetal(const float4& n, const float4& k, float roughness) {
    Material m = matDefault();
    m.type = MAT_METAL; m.hasTexture = false; m.eta = n; m.k = k; m.roughness = roughness;
    m.albedo = f4(1.0f); m.metallic = 1.0f; m.boundary = false; m.absorption = f4(); m.thinWalled = false; m.isSpecular = false;
    return m;
}
static Material matSmoothDielectric(float ior, const float4& k, int pri) {
    Material m = matDefault();
    m.type = MAT_SMOOTHDIELECTRIC; m.hasTexture = false; m.ior = ior; m.albedo = f4(1.0f);
    m.priority = pri; m.isSpecular = true; m.boundary = true; m.absorption = k; m.thinWalled = false;
    return m;
}
static Material matLeaf(int sInd, int w, int h, float ior, float roughness, float4 albedo, float transmission) {
    Material m = matDefault();
    m.type = MAT_LEAF; m.hasTexture = true; m.hasTransMap = false; m.ior = ior; m.roughness = roughness;
    m.albedo = albedo; m.transmission = transmission; m.boundary = false;
    m.startInd = sInd; m.width = w; m.height = h; m.thinWalled = true; m.isSpecular = false;
    return m;
}
static Material matMirror() {
    Material m = matDefault();
    m.type = MAT_DELTAMIRROR; m.hasTexture = false; m.hasTransMap = false; m.isSpecular = true;
    return m;
}

// imageUtil.cu:144-195 loadBMPToImage(path, isData=false): 24-bit BMP, rows read straight after the
// 54 header bytes (bfOffBits is ignored), each row padded to 4 bytes, y flipped, channels /255
// then pow(., 2.2) with the host libm, alpha 1. Missing / non-BMP / non-24-bit file -> 0x0 image.
static bool loadBMPToImage(const std::string& filename, std::vector<float4>& pixels, int& width, int& height) {
    width = height = 0;
    std::ifstream in(filename, std::ios::binary);
    if (!in.is_open()) return false;
    unsigned char hd[54];
    in.read(reinterpret_cast<char*>(hd), 54);
    if (!in || hd[0] != 'B' || hd[1] != 'M') return false;
    uint16_t bpp; std::memcpy(&bpp, hd + 28, 2);
    if (bpp != 24) return false;
    int32_t w, h; std::memcpy(&w, hd + 18, 4); std::memcpy(&h, hd + 22, 4);
    if (w <= 0 || h <= 0) return false;
    int rowSize = (3 * w + 3) & (~3);
    std::vector<unsigned char> row(rowSize);
    pixels.assign((size_t)w * h, f4());
    for (int y = 0; y < h; y++) {
        in.read(reinterpret_cast<char*>(row.data()), rowSize);
        for (int x = 0; x < w; x++) {
            float b = row[x * 3 + 0] / 255.0f, g = row[x * 3 + 1] / 255.0f, r = row[x * 3 + 2] / 255.0f;
            r = powf(r, 2.2f); g = powf(g, 2.2f); b = powf(b, 2.2f);
            pixels[(size_t)(h - 1 - y) * w + x] = f4(r, g, b, 1.0f);
        }
    }
    width = w; height = h;
    return true;
}

// main.cu:364-391: the four fixed textures, concatenated; a missing file contributes a 0x0 image.
void loadTextures(Scene& sc, const std::string& baseDir, int startIndices[4], int widths[4], int heights[4]) {
    const char* names[4] = {"textures/enkidutexture.bmp", "textures/enkiduchibitexture.bmp", "textures/leaftex2.bmp", "textures/leafautumn.bmp"};
    sc.textures.clear();
    int cur = 0;
    for (int i = 0; i < 4; i++) {
        std::vector<float4> px; int w, h;
        loadBMPToImage((baseDir.empty() ? std::string() : baseDir + "/") + names[i], px, w, h);
        sc.textures.insert(sc.textures.end(), px.begin(), px.end());
        widths[i] = w; heights[i] = h; startIndices[i] = cur;
        cur += w * h;
    }
}

// main.cu:397-467.
void buildMaterialTable(Scene& sc, const int startIndices[4], const int widths[4], const int heights[4]) {
    Material lambertTextured = matDiffuseTextured(startIndices[0], widths[0], heights[0]);
    Material lambert2Textured = matDiffuseTextured(startIndices[1], widths[1], heights[1]);
    Material lambertBlue = matDiffuse(f4(0.4f, 0.4f, 0.8f));
    Material lambertGrey = matDiffuse(f4(0.8f, 0.8f, 0.8f));
    Material lambertWhite = matDiffuse(f4(0.9f, 0.9f, 0.9f));
    Material lambertGreen = matDiffuse(f4(0.2f, 0.6f, 0.6f));
    Material lambertRed = matDiffuse(f4(0.90f, 0.1f, 0.1f));
    Material lambertVeryGreen = matDiffuse(f4(0.1f, 0.9f, 0.1f));
    Material lambertBLACK = matDiffuse(f4(0.0f, 0.0f, 0.0f));
    Material lambert95 = matDiffuse(f4(0.95f, 0.95f, 0.95f));
    Material lambert50 = matDiffuse(f4(0.5f, 0.5f, 0.5f));
    float4 eta_steel = f4(0.14f, 0.16f, 0.13f, 1.0f);
    float4 eta_gold = f4(0.17f, 0.35f, 1.5f);
    Material gold = matMetal(eta_gold, eta_gold, 0.05f);      // main.cu:419 passes eta as k
    Material steel = matMetal(eta_steel, eta_steel, 0.15f);   // main.cu:420
    Material glass = matSmoothDielectric(1.5f, f4(0.0f), 1);
    Material diamond = matSmoothDielectric(2.42f, f4(0.0f), 1);
    Material water = matSmoothDielectric(1.333f, f4(), 2);
    Material tea = matSmoothDielectric(1.333f, 2.5f * f4(0.180f, 1.5f, 2.996f), 2);
    Material ice = matSmoothDielectric(1.31f, f4(0.2f), 0);
    Material air = matSmoothDielectric(1.0f, f4(0.0f), 99);
    Material leaf = matLeaf(startIndices[2], widths[2], heights[2], 1.5f, 0.10f, f4(0.22f, 0.75f, 0.28f), 0.15f);
    Material leafAutumn = matLeaf(startIndices[3], widths[3], heights[3], 1.5f, 0.8f, f4(0.22f, 0.75f, 0.28f), 0.6f);
    Material leafStem = matDiffuse(f4(0.90f, 0.9f, 0.83f));
    Material sky = matDiffuse(f4(0.4f, 0.4f, 1.00f));
    Material mirror = matMirror();

    std::vector<Material>& mats = sc.mats;
    mats.clear();
    mats.push_back(air);               // 0
    mats.push_back(lambertBlue);       // 1
    mats.push_back(lambertWhite);      // 2
    mats.push_back(lambertGreen);      // 3
    mats.push_back(gold);              // 4
    mats.push_back(glass);             // 5
    mats.push_back(lambertRed);        // 6
    mats.push_back(steel);             // 7
    mats.push_back(tea);               // 8
    mats.push_back(ice);               // 9
    mats.push_back(water);             // 10
    mats.push_back(lambertTextured);   // 11
    mats.push_back(lambert2Textured);  // 12
    mats.push_back(leaf);              // 13
    mats.push_back(leafStem);          // 14
    mats.push_back(sky);               // 15
    mats.push_back(leafAutumn);        // 16
    mats.push_back(lambertGrey);       // 17
    mats.push_back(diamond);           // 18
    mats.push_back(mirror);            // 19
    mats.push_back(lambertBLACK);      // 20
    mats.push_back(lambert95);         // 21
    mats.push_back(lambert50);         // 22
    mats.push_back(lambertVeryGreen);  // 23
}

// ---- BVH, main.cu:20-233 -----------------------------------------------------------------
struct BuildCtx {
    std::vector<BVHnode>* nodes; std::vector<int>* indices;
    std::vector<float4> centroids, mins, maxes;
    int maxLeafSize, largestLeaf, backup;
};

// main.cu:20-47 (libm fminf/fmaxf on finite values)
static void computeInfoForBVH(const Scene& sc, BuildCtx& c) {
    for (size_t i = 0; i < sc.mesh.size(); i++) {
        const Triangle& tri = sc.mesh[i];
        float4 a = sc.points[tri.aInd], b = sc.points[tri.bInd], cc = sc.points[tri.cInd];
        c.centroids.push_back(f4((a.x + b.x + cc.x) / 3.0f, (a.y + b.y + cc.y) / 3.0f, (a.z + b.z + cc.z) / 3.0f));
        float4 minPos = f4(fminf(fminf(a.x, b.x), cc.x), fminf(fminf(a.y, b.y), cc.y), fminf(fminf(a.z, b.z), cc.z)) - f4(0.000001f);
        c.mins.push_back(minPos);
        float4 maxPos = f4(fmaxf(fmaxf(a.x, b.x), cc.x), fmaxf(fmaxf(a.y, b.y), cc.y), fmaxf(fmaxf(a.z, b.z), cc.z)) + f4(0.000001f);
        c.maxes.push_back(maxPos);
    }
}

// main.cu:49-62
static int partitionPrimitives(BuildCtx& c, int start, int end, int axis, float splitPos) {
    std::vector<int>& indices = *c.indices;
    int mid = start;
    for (int i = start; i < end; i++) {
        float4 cen = c.centroids[indices[i]];
        if (getFloat4Component(cen, axis) < splitPos) { std::swap(indices[i], indices[mid]); mid++; }
    }
    return mid;
}

// main.cu:64-131. 12 buckets over the node bounds on `axis`. Quirk kept: the right side
// starts from bucket i AND loops j from i, so bucket i's count is added twice (:102-109).
static void SAH(BuildCtx& c, int start, int end, int axis, float4 minBound, float4 maxBound, float& splitPos, float& minCost) {
    std::vector<int>& indices = *c.indices;
    const int numBuckets = 12;
    struct Bucket { float4 min, max; int count; };
    Bucket buckets[numBuckets];
    for (int i = 0; i < numBuckets; i++) { buckets[i].min = f4(FLT_MAX); buckets[i].max = f4(-FLT_MAX); buckets[i].count = 0; }
    for (int i = start; i < end; i++) {
        int idx = indices[i];
        float cen = getFloat4Component(c.centroids[idx], axis);
        float denom = getFloat4Component(maxBound, axis) - getFloat4Component(minBound, axis);
        float q = numBuckets * (cen - getFloat4Component(minBound, axis)) / denom;
        int b = (q == q && fabsf(q) < 1e9f) ? (int)q : 0;      // int(NaN/inf) is UB in the reference; defined 0
        b = b < 0 ? 0 : (b > numBuckets - 1 ? numBuckets - 1 : b);
        buckets[b].count++;
        buckets[b].min = fminf4(buckets[b].min, c.mins[idx]);
        buckets[b].max = fmaxf4(buckets[b].max, c.maxes[idx]);
    }
    minCost = FLT_MAX;
    int bestSplit = -1;
    for (int i = 1; i < numBuckets; i++) {
        float4 leftMin = buckets[0].min, leftMax = buckets[0].max;
        int leftCount = buckets[0].count;
        for (int j = 1; j < i; j++) { leftMin = fminf4(leftMin, buckets[j].min); leftMax = fmaxf4(leftMax, buckets[j].max); leftCount += buckets[j].count; }
        float4 rightMin = buckets[i].min, rightMax = buckets[i].max;
        int rightCount = buckets[i].count;
        for (int j = i; j < numBuckets; j++) { rightMin = fminf4(rightMin, buckets[j].min); rightMax = fmaxf4(rightMax, buckets[j].max); rightCount += buckets[j].count; }
        float cost = 1.0f + (leftCount * surfaceArea(leftMin, leftMax) + rightCount * surfaceArea(rightMin, rightMax)) / surfaceArea(minBound, maxBound);
        if (cost < minCost && (leftCount > 0 && rightCount > 0)) { minCost = cost; bestSplit = i; }
    }
    if (bestSplit == -1) {
        // main.cu:119-128 uses std::nth_element (STL-specific permutation). SURVEY App. D:
        // replaced by a fully specified order — sort the range by (centroid[axis], index).
        int mid = (start + end) / 2;
        std::sort(indices.begin() + start, indices.begin() + end, [&](int a, int b) {
            float ca = getFloat4Component(c.centroids[a], axis), cb = getFloat4Component(c.centroids[b], axis);
            if (ca < cb) return true;
            if (cb < ca) return false;
            return a < b;
        });
        splitPos = getFloat4Component(c.centroids[indices[mid]], axis);
    } else {
        splitPos = getFloat4Component(minBound, axis) + (getFloat4Component(maxBound, axis) - getFloat4Component(minBound, axis)) * (float(bestSplit) / float(numBuckets));
    }
}

// main.cu:133-233
static int buildBVH(BuildCtx& c, int start, int end) {
    std::vector<BVHnode>& nodes = *c.nodes;
    std::vector<int>& indices = *c.indices;
    int nodeIndex = (int)nodes.size();
    nodes.push_back(BVHnode());
    float4 minBound = c.mins[indices[start]];
    float4 maxBound = c.maxes[indices[start]];
    for (int i = start; i < end; i++) {
        int idx = indices[i];
        minBound = fminf4(minBound, c.mins[idx]);
        maxBound = fmaxf4(maxBound, c.maxes[idx]);
    }
    nodes[nodeIndex].aabbMIN = minBound;
    nodes[nodeIndex].aabbMAX = maxBound;
    int primCount = end - start;
    if (primCount <= c.maxLeafSize) {
        nodes[nodeIndex].first = start; nodes[nodeIndex].primCount = primCount;
        nodes[nodeIndex].left = nodes[nodeIndex].right = -1;
        c.largestLeaf = std::max(primCount, c.largestLeaf);
        return nodeIndex;
    }
    float xdiff = maxBound.x - minBound.x, ydiff = maxBound.y - minBound.y, zdiff = maxBound.z - minBound.z;
    int axis = 0;
    if (ydiff > xdiff && ydiff > zdiff) axis = 1;
    else if (zdiff > xdiff && zdiff > ydiff) axis = 2;

    float splitPos, cost = 0;
    SAH(c, start, end, axis, minBound, maxBound, splitPos, cost);

    int mid = start;
    int numLeft = 0;
    for (int i = start; i < end; i++) if (getFloat4Component(c.centroids[indices[i]], axis) < splitPos) numLeft++;
    if (numLeft > 0 && numLeft < (primCount - 1)) {
        mid = partitionPrimitives(c, start, end, axis, splitPos);
    } else {
        float sum = 0.0f;
        c.backup++;
        for (int i = start; i < end; i++) sum += getFloat4Component(c.centroids[indices[i]], axis);
        splitPos = sum / primCount;
    }
    numLeft = 0;
    for (int i = start; i < end; i++) if (getFloat4Component(c.centroids[indices[i]], axis) < splitPos) numLeft++;
    if (numLeft > 0 && numLeft < (primCount - 1)) {
        mid = partitionPrimitives(c, start, end, axis, splitPos);
    } else {
        nodes[nodeIndex].first = start; nodes[nodeIndex].primCount = primCount;
        nodes[nodeIndex].left = nodes[nodeIndex].right = -1;
        c.largestLeaf = std::max(primCount, c.largestLeaf);
        return nodeIndex;
    }
    int l = buildBVH(c, start, mid);
    nodes[nodeIndex].left = l;
    int r = buildBVH(c, mid, end);
    nodes[nodeIndex].right = r;
    nodes[nodeIndex].primCount = 0;
    nodes[nodeIndex].first = -1;
    return nodeIndex;
}

static int treeDepth(const std::vector<BVHnode>& n, int i) {
    // iterative to survive degenerate chains
    int best = 0;
    std::vector<std::pair<int, int>> st; st.push_back({i, 1});
    while (!st.empty()) {
        auto [idx, d] = st.back(); st.pop_back();
        best = std::max(best, d);
        if (n[idx].primCount > 0) continue;
        if (n[idx].left >= 0) st.push_back({n[idx].left, d + 1});
        if (n[idx].right >= 0) st.push_back({n[idx].right, d + 1});
    }
    return best;
}

// main.cu:502-530
void buildSceneBVH(Scene& sc, int maxLeafSize) {
    sc.bvh.clear();
    sc.indices.resize(sc.mesh.size());
    for (size_t i = 0; i < sc.mesh.size(); i++) sc.indices[i] = (int)i;
    if (sc.mesh.empty()) return;
    BuildCtx c;
    c.nodes = &sc.bvh; c.indices = &sc.indices; c.maxLeafSize = maxLeafSize; c.largestLeaf = 0; c.backup = 0;
    computeInfoForBVH(sc, c);
    buildBVH(c, 0, (int)sc.mesh.size());
    sc.largestLeaf = c.largestLeaf;
    sc.backupCount = c.backup;
    sc.maxDepthOfTree = treeDepth(sc.bvh, 0);
}

// ---- camera, objects.cuh:221-264, 309-325 (host libm tanf/cosf/sinf) ----------------------
static float4 rotateX(const float4& v, float angle) { float c = cosf(angle), s = sinf(angle); return f4(v.x, v.y * c - v.z * s, v.y * s + v.z * c); }   // util.cuh:237
static float4 rotateY(const float4& v, float angle) { float c = cosf(angle), s = sinf(angle); return f4(v.x * c + v.z * s, v.y, -v.x * s + v.z * c); }  // util.cuh:247
static float4 rotateZ(const float4& v, float angle) { float c = cosf(angle), s = sinf(angle); return f4(v.x * c - v.y * s, v.x * s + v.y * c, v.z); }   // util.cuh:257

static void preCompute(Camera& c) {
    c.forward = normalize(rotateZ(rotateY(rotateX(f4(0.0f, 0.0f, -1.0f, 0.0f), c.xRot), c.yRot), c.zRot));
    c.right = normalize(rotateZ(rotateY(rotateX(f4(1.0f, 0.0f, 0.0f, 0.0f), c.xRot), c.yRot), c.zRot));
    c.up = normalize(rotateZ(rotateY(rotateX(f4(0.0f, 1.0f, 0.0f, 0.0f), c.xRot), c.yRot), c.zRot));
}

Camera cameraPinhole(const float4& origin, int w, int h, float xR, float yR, float zR, float FOV, float aajitter) {
    Camera c; std::memset(&c, 0, sizeof(c));
    c.w = w; c.h = h; c.cameraOrigin = origin;
    c.fovScale = tanf((FOV * 0.5f) * (3.141592f / 180.0f));
    c.xRot = xR * (3.14159265f / 180.0f); c.yRot = yR * (3.14159265f / 180.0f); c.zRot = zR * (3.14159265f / 180.0f);
    c.aperture = 0.000001f;            // objects.cuh:234 — a "pinhole" still samples the lens
    c.focalDist = 1.0f / FOV;          // objects.cuh:235
    c.antiAliasJitterDist = aajitter;
    preCompute(c);
    return c;
}

Camera cameraNotPinhole(const float4& origin, int w, int h, float xR, float yR, float zR, float FOV, float aperture, float focalDist, float aajitter) {
    Camera c; std::memset(&c, 0, sizeof(c));
    c.w = w; c.h = h; c.cameraOrigin = origin;
    c.fovScale = tanf((FOV * 0.5f) * (3.141592f / 180.0f));
    c.xRot = xR * (3.14159265f / 180.0f); c.yRot = yR * (3.14159265f / 180.0f); c.zRot = zR * (3.14159265f / 180.0f);
    c.aperture = aperture; c.focalDist = focalDist; c.antiAliasJitterDist = aajitter;
    preCompute(c);
    return c;
}

// main.cu:235-557 minus device uploads.
bool loadSceneFromConfig(const std::string& configPath, const std::string& baseDir, int renderNumber,
                         RenderConfig& cfg, Scene& sc, Camera& cam) {
    if (!loadConfig(configPath, cfg)) return false;
    if (cfg.pinholeCamera)
        cam = cameraPinhole(cfg.camPos, cfg.width, cfg.height, cfg.camRot.x, cfg.camRot.y, cfg.camRot.z, cfg.camFov);
    else
        cam = cameraNotPinhole(cfg.camPos, cfg.width, cfg.height, cfg.camRot.x, cfg.camRot.y, cfg.camRot.z, cfg.camFov, cfg.camApeture, cfg.camFocalDist);
    int starts[4], tw[4], th[4];
    loadTextures(sc, baseDir, starts, tw, th);
    buildMaterialTable(sc, starts, tw, th);
    for (const MeshConfig& m : cfg.meshes) {
        std::string p = (baseDir.empty() || (!m.path.empty() && m.path[0] == '/')) ? m.path : baseDir + "/" + m.path;
        float4 e = m.emissionMultiplier * m.emissionColor;
        if (lengthSquared(m.emissionColor) > 0.0f)
            readObjSimple(p, sc, e, m.materialID, f4(0.0f, -0.01f * renderNumber, 0.0f));   // main.cu:476-478
        else
            readObjSimple(p, sc, e, m.materialID, f4(0.0f));
    }
    if (sc.mesh.empty()) { std::cerr << "Error: No triangles loaded." << std::endl; return false; }
    buildSceneBVH(sc, cfg.bvhLeafSize);
    return true;
}

}  // namespace oracle
