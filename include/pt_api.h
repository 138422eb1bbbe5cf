/* pt_api.h — C ABI of the MI355X-native unidirectional path tracer (libptamd.so).
 *
 * Drop-in boundary for ONE hot path of DanielQ-51/cudapathtracer: the integrator launchers
 *     launch_unidirectional / launch_naive_unidirectional        (deviceCode.cuh:8-12,
 *                                                                 bodies deviceCode.cu:544-620, 207-283)
 * and the kernels under them (initRNG deviceCode.cu:53-61, Li_unidirectional :285-542,
 * Li_naive_unidirectional :158-205). Everything here is plain C: pointers, sizes, PODs whose
 * byte layouts equal the reference's CUDA structs, no torch / HIP types.
 *
 * The reference passes eleven raw device pointers per launch; here the scene arguments are
 * bundled once into an opaque `pt_scene` (which re-packs them for gfx950) and the
 * launchers take that handle. Errors: every entry point returns 0 on success or a negative
 * code, never throws or exits (the reference's launchers return void and print,
 * deviceCode.cu:611-619); pt_last_error() holds the message for the calling thread.
 *
 * `novum_*` entry points are the host side the reference keeps in main.cu/objects.cuh (config
 * parser, OBJ reader, SAH BVH builder, material table, camera, finalise) restated in plain C++.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_API_VERSION 1
#define PT_TILE_DIM 8          /* one wave64 renders one 8x8-pixel tile */
#define PT_TILE_PIXELS 64

/* ---- PODs with the reference's layouts ------------------------------------------------- */
typedef struct pt_float4 { float x, y, z, w; } __attribute__((aligned(16))) pt_float4;   /* CUDA float4 */
typedef struct pt_float2 { float x, y; } __attribute__((aligned(8))) pt_float2;           /* CUDA float2 */

/* objects.cuh:12-20 — 48 B. Leaf: left = right = -1, first/primCount index BVHindices.
 * Internal: primCount = 0, first = -1. Root is node 0 (main.cu:133-233). */
typedef struct pt_bvh_node {
    pt_float4 aabbMIN, aabbMAX;
    int32_t left, right, first, primCount;
} pt_bvh_node;

/* objects.cuh:159-172 — 80 B. lightInd = index into the light list or -51 (main.cu:1054-1056). */
typedef struct pt_triangle {
    int32_t aInd, bInd, cInd;
    int32_t naInd, nbInd, ncInd;
    int32_t uvaInd, uvbInd, uvcInd;
    int32_t materialID;
    pt_float4 emission;
    int32_t lightInd, triInd;
} pt_triangle;

/* objects.cuh:605-638 — 176 B. `type` is MaterialType (objects.cuh:595-603). */
typedef struct pt_material {
    uint8_t hasTexture; int32_t startInd, width, height;
    uint8_t hasTransMap; int32_t tstartInd, twidth, theight;
    int32_t type;
    pt_float4 albedo;
    float roughness;
    pt_float4 eta, k;
    float ior, metallic, specular, transmission;
    uint8_t isSpecular, boundary, thinWalled;
    pt_float4 absorption;
    int32_t priority;
} pt_material;

/* objects.cuh:199-219 — 112 B, passed by value to the reference's kernels. */
typedef struct pt_camera {
    pt_float4 cameraOrigin;
    int32_t w, h;
    float xRot, yRot, zRot;
    float aperture, focalDist, fovScale;
    float antiAliasJitterDist;
    pt_float4 forward, right, up;
} pt_camera;

enum { PT_MAT_DIFFUSE = 0, PT_MAT_METAL = 1, PT_MAT_SMOOTHDIELECTRIC = 2, PT_MAT_MICROFACETDIELECTRIC = 3,
       PT_MAT_LEAF = 4, PT_MAT_FLOWER = 5, PT_MAT_DELTAMIRROR = 6 };          /* objects.cuh:595-603 */
enum { PT_UNIDIRECTIONAL = 0, PT_NAIVE_UNIDIRECTIONAL = 2 };                  /* objects.cuh:570-576 */

/* What initRender uploads before the launch (main.cu:469-557), as HOST arrays. */
typedef struct pt_scene_desc {
    const pt_float4* positions; int32_t n_positions;       /* Vertices.positions */
    const pt_float4* normals;   int32_t n_normals;         /* Vertices.normals   */
    const pt_float2* uvs;       int32_t n_uvs;             /* Vertices.uvs       */
    const pt_triangle* triangles; int32_t n_triangles;     /* `scene`            */
    const pt_triangle* lights;    int32_t n_lights;        /* `lights` (copies of the emissive triangles) */
    const pt_bvh_node* bvh;       int32_t n_nodes;         /* `BVH`              */
    const int32_t* bvh_indices;                            /* `BVHindices`, n_triangles entries */
    const pt_material* materials; int32_t n_materials;     /* `materials`        */
    const pt_float4* textures;    int32_t n_texels;        /* `textures`         */
} pt_scene_desc;

typedef struct pt_scene pt_scene;      /* opaque: device-resident, re-packed scene */

/* A rank's share of the framebuffer: tiles {first + i*stride : 0 <= i < count} of the row-major
 * grid of ceil(w/8) x ceil(h/8) 8x8-pixel tiles. NULL means every tile. */
typedef struct pt_tile_range { int32_t first, stride, count; } pt_tile_range;

/* Work counters of SURVEY.md §8(d), summed over every render since pt_reset_counters. */
typedef struct pt_counters {
    uint64_t rays_closest, rays_shadow, node_pops, box_tests, tri_tests, hits, rng_draws, iterations;
} pt_counters;

/* ---- library ------------------------------------------------------------------------- */
int pt_api_version(void);
const char* pt_last_error(void);
int pt_device_count(void);                 /* number of HIP devices, < 0 on error */

/* ---- scene --------------------------------------------------------------------------- */
/* Re-packs and uploads the scene to the CURRENT HIP device. NULL on error. Replaces the
 * cudaMalloc/cudaMemcpy block main.cu:469-557. */
pt_scene* pt_scene_create(const pt_scene_desc* desc);
void pt_scene_destroy(pt_scene* scene);

/* ---- launchers (the hot path) -------------------------------------------------------- */
/* launch_unidirectional (integrator 0, useMIS as at main.cu:565) / launch_naive_unidirectional
 * (integrator 2): seeds one XORWOW stream per pixel keyed by the GLOBAL index y*w+x
 * (deviceCode.cu:59-60), runs `spp` samples per pixel with the stream continuing across
 * samples, and ADDS the radiance sum into `out_rgba_sum` (w*h float4, row-major, y = 0 is the
 * bottom row; `colors[pixelIdx] += Li`, deviceCode.cu:540). Host-buffer, blocking form. Only
 * pixels of tiles in `tiles` are touched. */
int pt_render(pt_scene* scene, const pt_camera* camera, int w, int h, int spp, int max_depth,
              int integrator, int use_mis, uint64_t seed, const pt_tile_range* tiles, float* out_rgba_sum);

/* Device-resident forms, asynchronous on `stream` (a hipStream_t, NULL = default stream).
 * d_tile_rgba: count*64 float4, tile-major ([local tile][ly*8+lx]); += semantics.
 * A pt_scene owns its work buffers (per-pixel RNG states, tile queue, traversal spill area): launches on ONE
 * scene must be ordered (same stream, or synchronised); different scenes are independent. */
int pt_render_tiles_device(pt_scene* scene, const pt_camera* camera, int w, int h, int spp, int max_depth,
                           int integrator, int use_mis, uint64_t seed, const pt_tile_range* tiles,
                           void* d_tile_rgba, int count_work, void* stream);
/* scan-line `colors` (w*h float4) <- tile-major buffer, for the tiles of `tiles`. */
int pt_untile_device(int w, int h, const pt_tile_range* tiles, const void* d_tile_rgba, void* d_colors, void* stream);
/* tile-major buffer <- scan-line `colors` (to continue an accumulation). */
int pt_tile_device(int w, int h, const pt_tile_range* tiles, const void* d_colors, void* d_tile_rgba, void* stream);

/* The reference's own call shape with the scene bundled (deviceCode.cuh:8-12): d_colors is the
 * zero-initialised DEVICE accumulator of main.cu:337-339; blocking; seed 103033
 * (deviceCode.cu:552); vertNum/triNum/lightNum live in the scene. */
int pt_launch_unidirectional(int maxDepth, pt_camera camera, pt_scene* scene, int numSample, int useMIS, int w, int h, void* d_colors);
int pt_launch_naive_unidirectional(int maxDepth, pt_camera camera, pt_scene* scene, int numSample, int useMIS, int w, int h, void* d_colors);

/* The launchers with the reference's progressive hook (the `elapsed >= saveIntervalSeconds` block
 * inside the sample loop, deviceCode.cu:574-604 / 237-267). The samples run in chunks of
 * chunk_spp; after each chunk d_colors holds the sum over samples_done samples and
 * progress(samples_done, user) runs on the calling thread (write a preview there — the reference
 * writes render.bmp + renderCSV.csv; this ABI does no file I/O). Per-pixel streams continue across
 * chunks: the final d_colors is bit-identical to pt_launch_[naive_]unidirectional. A non-zero
 * return from progress ends the render after that chunk. integrator: 0 or 2. */
typedef int (*pt_progress_fn)(int samples_done, void* user);
int pt_launch_progressive(int integrator, int maxDepth, pt_camera camera, pt_scene* scene, int numSample, int useMIS, int w, int h,
                          void* d_colors, int chunk_spp, pt_progress_fn progress, void* user);

/* The same image from the COUNTING instantiations of the kernels (stack walk everywhere, no time slices), plus the
 * per-pixel counters (w*h x 8 uint32: rays_closest, rays_shadow, node_pops, box_tests, tri_tests, hits, rng_draws,
 * iterations; out_counters may be NULL) and the totals of pt_get_counters, for parity checks. pt_render itself runs
 * the kernels that bench.py times and leaves the counters alone. */
int pt_render_counted(pt_scene* scene, const pt_camera* camera, int w, int h, int spp, int max_depth,
                      int integrator, int use_mis, uint64_t seed, const pt_tile_range* tiles,
                      float* out_rgba_sum, uint32_t* out_counters);

/* Kernel organisation used by every later render of this scene: 0 = megakernel (default: one wave
 * per 8x8 tile, all samples and bounces inside one launch), 1 = wavefront (stream-compacted: path
 * state in HBM, one logic + one traversal launch per bounce; BASELINE config 5's divergence A/B).
 * Results are bit-identical. */
int pt_set_variant(pt_scene* scene, int variant);

int pt_get_counters(pt_scene* scene, pt_counters* out);
int pt_reset_counters(pt_scene* scene);
/* Device time (ms) of the most recent megakernel launch on this scene, from HIP events recorded
 * on the launch stream around that kernel alone; waits for the launch to finish. Returns -1 (and
 * sets pt_last_error) if that launch did not complete its frame — the tile queue's bounded waits
 * ran out. The blocking launchers return -4 in that case themselves; a caller of the asynchronous
 * pt_render_tiles_device MUST check here (or through any later blocking launcher) before it uses
 * the tile buffer. */
float pt_last_kernel_ms(pt_scene* scene);
/* Tile hand-overs of the last megakernel launch (call after it has completed): how many times a wave yielded its tile at
 * the end of a time slice for another wave to continue (0 without time slices). For tests and scheduling measurements. */
int pt_last_tile_handovers(pt_scene* scene);
/* How many launches of this scene had their queue waiters give up — no tile finished or handed over for "queue_timeout_ms"
 * (option, default 30 000) — although every tile was finished in the end. Not an error: waiters hold no tile, the frame is
 * complete and exact; it says the device stalled (seen with several persistent kernels co-resident on one device). A launch
 * that ends with unfinished tiles IS an error (-4). */
int pt_queue_stalls(pt_scene* scene);
/* Debug: the 16 header words of the tile queue after the last queued launch of this scene (1: copied, 0: the last launch used no
 * queue). [0] pops claimed, [1] pushes claimed, [2] tiles finished, [3] stall / error bits (1: a waiter recorded a stall and kept
 * waiting, 2: a waiter gave up for good, 4: a push found no slot), [4] / [5] the issue-priority steering's sums (zero once the kernel
 * has ended), [6] / [7] what the first stalled waiter saw, [8..9] the wait bound in ticks of the 100 MHz steady counter. */
int pt_debug_queue_header(pt_scene* scene, int* out16);
/* Which instantiation the launcher picks for this scene: bit 0 = ONCHIP (whole packed scene in the LDS cache),
 * bit 1 = persistent waves on the tile queue, bit 2 = time slices on, bit 3 = the 6-waves-per-SIMD kernel for scenes in HBM
 * (as used by the last launch; it needs enough tiles), bit 4 = opt-in culling, bit 5 = the last launch used a REFILL
 * instantiation (scenes in HBM: finished lanes shade and return while the others keep tracing), bit 6 = it used the FLAT
 * closest-hit traversal (LDS-resident scenes with at most 64 nodes and triangles), bit 7 = it used the SIMPLE bounce
 * (every triangle an untextured MAT_DIFFUSE: one arm per dispatcher, no medium stack), bit 8 = the pair form of FLAT
 * (shadow + extension ray in one pass), bit 9 = the FLAT launch decided the visited leaves from the leaves' own boxes (the
 * scene's boxes are finite and nested, checked at pt_scene_create; a caller's loose or refit tree takes the lockstep walk
 * over the boxes as given), bit 10 = it used the LEAN generic bounce. For labelling measurements. */
int pt_scene_flags(pt_scene* scene);
/* Opt-in (default off): skip BVH children whose box lies beyond the best hit so far / beyond a shadow ray's max_t.
 * The reference has no such test and its results are the contract, so the default kernels do not have it either: a
 * triangle inside a skipped box can still produce a smaller t (different roundings, grazing incidence), and one such
 * hit shifts the pixel's whole RNG stream. Measured on the 263 k-triangle scene: 13 of 2 073 600 pixels differ after
 * 4.6e9 rays, at 1.2-1.6x the speed. A renderer's trade-off, not the reference's image. Applies to the
 * kernel for scenes that do not fit the LDS cache; flag bit 4 of pt_scene_flags reports it. */
int pt_set_culling(pt_scene* scene, int on);
/* Per-scene kernel-selection and scheduling options, by name. The library reads NO environment variable: what a
 * process renders cannot be steered from its environment. Only "culling" (= pt_set_culling) can reach the image; all
 * other options choose between instantiations / schedules whose results are bit-identical (tests/test_gpu_parity.py
 * drives every one of them against the oracle). They exist for A/B measurements and for the tests.
 *   "flat" 0|1            FLAT closest-hit traversal for LDS-resident scenes of at most 128 nodes and triangles (1; 2 = 1)
 *   "wf_wide_wg" 0|1|2    wavefront variant, scenes in HBM: the trace kernel in 16-wave workgroups sharing 48 KB of the tree top (1: when
 *                         the launch has enough paths to fill them, 2: always)
 *   "leaf_boxes" 0|1      FLAT kernels test each leaf's own box instead of walking the nodes in lockstep (1)
 *   "flat2" 0|1           FLAT scenes (<= 64 triangles, no MAT_LEAF triangle), MIS integrator: shadow ray and next extension
 *                         ray in one FLAT pass (1)
 *   "simple" 0|1          with FLAT: the diffuse-only bounce for scenes whose triangles are all untextured MAT_DIFFUSE (1)
 *   "lean" 0|1            the generic bounce without its leaf arms and texture fetches for scenes that have neither a MAT_LEAF
 *                         triangle nor a textured material — glass, mirrors, metals (1)
 *   "onchip" 0|1          LDS-resident instantiation when the scene fits (1)
 *   "waves_hbm" 0|1|2     the 6-waves-per-SIMD kernel for scenes in HBM: never / when the launch has enough tiles / always (1)
 *   "refill" 0|1          resumable traversal for scenes in HBM (1)
 *   "refill_keep", "node_keep", "tri_keep" 0..15   loop-exit thresholds in sixteenths (6, 10, 8; re-swept in round 3: profiles/r03_sweep_keep.log — while "refill_keep" is unset the 4-wave kernel of small shares uses 10)
 *   "slice_iters" n       bounce iterations a wave keeps a tile before it queues it again, 0 = until finished (512)
 *   "slice_always" 0|1    time slices from the first tile on (1)
 *   "sched_mask" 2^k-1    a wave looks at the queue every sched_mask + 1 iterations (31)
 *   "lpt_prio" 0|1|2      issue-priority steering: off / once no fresh tile is left / always (2)
 *   "persistent" 0|1      persistent waves on the tile queue (1)
 *   "queue_timeout_ms" n  how long a wait on the tile queue may see no progress before the waiters leave (30 000)
 * Experimental options — variants that were built, proven bit-identical and measured SLOWER. The default
 * library does not contain their kernels (pt_has_experimental() == 0) and accepts only their "off" value, returning -3
 * otherwise; `make -C cudapathtracer_amd/csrc EXPERIMENTAL=1` builds them for the A/B:
 *   "wide" 0|1 (4-wide collapsed tree), "compact" 0|1 (32-byte quantised nodes), "spec" 1|2 (speculative descent, with
 *   -DPT_SPEC=1), "defer_shadow" 0|1 (pair walk in the 4-wave kernel), "xcd_bands" 0|1 (one band of tiles per XCD),
 *   "refill" 2 (resumable traversal for LDS-resident scenes too).
 * Returns 0, or < 0 for an unknown name / a value out of range. */
int pt_set_option(pt_scene* scene, const char* name, int value);
int pt_get_option(pt_scene* scene, const char* name, int* value);
int pt_has_experimental(void);
/* Eight more sums since the last pt_reset_counters. Normal build: out8[0] = internal-node fetches of counting launches
 * that went to global memory (node index beyond the LDS scene cache) — with tri_tests, the L1 line-access count behind
 * bench.py's roofline for scenes in HBM; the rest zero. Diagnostic builds:
 * -DPT_STAMPS: s_memtime spent in regeneration, closest-hit traversal, bounce logic (incl. the shadow ray), then the
 * sum of wave lifetimes, ~(earliest start) and the latest end on the 100 MHz wall clock, 0, 0 (tools/stamps.py).
 * -DPT_UTIL (counting launches): {trips of a wave, trips summed over its lanes} through the node loop and the
 * triangle loop of the closest-hit traversal, then of the shadow traversal (tools/lane_util.py). */
int pt_debug_stamps(pt_scene* scene, unsigned long long* out8);

/* ---- multi-GPU: the framebuffer sharded by screen tile over the GPUs of one node -------------
 * The reference is single-GPU (one launch_unidirectional per frame, main.cu:565). A pixel depends only on (scene,
 * camera, global pixel index, seed), so device r of N renders the 8x8 tiles {t : t mod N == r} of a replicated scene
 * — no data-path collective — and ONE gather (RCCL send/recv over xGMI, or hipMemcpyPeerAsync) brings the tile
 * buffers to device 0, which de-interleaves them. One host thread per device inside the call; blocking; the result
 * is bit-identical to pt_render for any N. This is what a C++ host (the reference's main.cu) binds to use 8 GPUs. */
#define PT_MULTI_MAX_DEVICES 16
typedef struct pt_multi pt_multi;          /* opaque: one scene replica, stream and communicator per device */
typedef struct pt_multi_stats {
    int32_t n_devices;
    int32_t gather;                        /* transport the gather used: 0 none (one device), 1 RCCL, 2 peer copies */
    float kernel_ms[PT_MULTI_MAX_DEVICES]; /* per device: its megakernel, HIP events on its stream */
    float render_ms, gather_ms, total_ms;  /* host wall clock: upload + render (max over devices), gather, whole call */
} pt_multi_stats;
/* Tile ownership of `rank` among `world` devices (interleaved). */
void pt_rank_tiles(int w, int h, int rank, int world, pt_tile_range* out);
/* Replicates the scene on n_devices HIP devices (device_ids NULL = 0 .. n_devices-1). NULL on error. */
pt_multi* pt_multi_create(const pt_scene_desc* desc, int n_devices, const int* device_ids);
void pt_multi_destroy(pt_multi* m);
/* "gather": 0 auto (RCCL, else peer copies), 1 RCCL, 2 hipMemcpyPeerAsync; "self_gather" 1: with ONE device, still send
 * the tile buffer through the collective (to itself) — a plumbing check for one-GPU machines; "same_device" 1 (default):
 * ranks that share a device (rehearsals only) issue to ONE stream of that device, so their persistent kernels — each
 * sized to fill the chip — run in stream order, 0: every rank its own stream, the kernels co-reside (both bit-identical;
 * the host threads run concurrently either way); any pt_set_option name: applied to every replica. */
int pt_multi_set_option(pt_multi* m, const char* name, int value);
int pt_multi_set_variant(pt_multi* m, int variant);
/* pt_render over all devices of `m`: host buffer in / out with `+=` semantics like pt_render. stats may be NULL. */
int pt_multi_render(pt_multi* m, const pt_camera* camera, int w, int h, int spp, int max_depth, int integrator, int use_mis,
                    uint64_t seed, float* out_rgba_sum, pt_multi_stats* stats);
/* One-shot form: create, render, destroy. */
int pt_render_multi(const pt_scene_desc* desc, int n_devices, const int* device_ids, const pt_camera* camera, int w, int h, int spp,
                    int max_depth, int integrator, int use_mis, uint64_t seed, float* out_rgba_sum, pt_multi_stats* stats);

/* ---- probes: single stages of the path on the GPU, for known-answer tests -------------- */
int pt_probe_rng(uint64_t seed, int n, const uint32_t* subsequences, int n_draws, uint32_t* out_state6, uint32_t* out_u32, float* out_uniform);
int pt_probe_math(int n, const float* x, float* out_sin, float* out_cos, float* out_exp, float* out_rsqrt, float* out_pow5);
/* All 2^32 binary32 inputs through the kernels' exact fast reciprocal (v_rcp_f32 + one Newton step inside 1e-12 <= |a| <= 1e30,
 * the IEEE division outside) against the IEEE division it stands for (`f = 1.0 / a`, integratorUtilities.cuh:22; 1 / dir, :50-55;
 * rsqrtf, util.cuh:129). out3[0] = inputs whose results differ (the arithmetic contract needs 0 on the device at hand),
 * out3[1] = inputs inside the fast range, out3[2] = inputs outside it where the bare fast sequence would be wrong (the reason
 * for the range guard). first_bad (may be NULL): the lowest differing bit pattern, 0xffffffff if none. ~0.1 s. */
int pt_probe_rcp_exhaustive(unsigned long long* out3, uint32_t* first_bad);
int pt_probe_camera_rays(const pt_camera* camera, uint64_t seed, int n, const int32_t* xy, float* out_rays6);
/* rays: n x 6 floats. out_i: n x 4 (valid, triIDX, materialID, backface);
 * out_f: n x 12 (t,u,v, point xyz, normal xyz, uv xy, 0); counters: summed over the n rays. */
int pt_probe_trace_closest(pt_scene* scene, int n, const float* rays6, int32_t* out_i, float* out_f, pt_counters* counters);
int pt_probe_trace_shadow(pt_scene* scene, int n, const float* rays6, const float* max_t, float* out_throughput3, pt_counters* counters);
/* sample_f_eval on stream (seed, subseq): out 8 floats (wo xyz, f xyz, pdf, draws) */
int pt_probe_bsdf_sample(pt_scene* scene, int n, const int32_t* material, const float* wi3, const int32_t* backface,
                         float etaI, float etaT, uint64_t seed, const uint32_t* subseq, float* out8);
/* f_eval + pdf_eval: out 4 floats (f xyz, pdf) */
int pt_probe_bsdf_eval(pt_scene* scene, int n, const int32_t* material, const float* wi3, const float* wo3,
                       float etaI, float etaT, float* out4);

/* ---- f-4: BVH build on the device --------------------------------------------------------
 * Replaces computeInfoForBVH + buildBVH (main.cu:20-233; call site main.cu:524-530: host vectors
 * `bvhvec` / `indvec` out of `points` / `mesh`). PT_BVH_REFERENCE_TREE: the nodes (pre-order, as
 * nodes.size() numbers them, main.cu:137) and the BVHindices permutation are byte-identical to the
 * host builder's. Host pointers in and out; nodes_out needs room for 2*n_triangles-1 nodes.
 * Returns the node count (> 0) or a negative error (pt_last_error()). No CPU fallback: without a
 * HIP device it fails; novum_bvh_build_host is the kept host builder on the same arrays. */
#define PT_BVH_REFERENCE_TREE 0
typedef struct pt_bvh_build_stats {
    int32_t n_nodes, largest_leaf, backups, depth;   /* what main.cu:537-538 prints */
    int32_t sort_fallbacks, levels;
    float device_ms;                                 /* HIP events around the build kernels */
    float total_ms;                                  /* wall clock incl. allocation, upload, download */
} pt_bvh_build_stats;
int pt_bvh_build_device(const pt_float4* positions, int n_positions, const pt_triangle* triangles, int n_triangles,
                        int max_leaf_size, int mode, pt_bvh_node* nodes_out, int nodes_capacity,
                        int32_t* indices_out, pt_bvh_build_stats* stats);
/* The same, without leaving the device: builds the reference tree from desc->positions / triangles (desc->bvh and
 * desc->bvh_indices are ignored and may be NULL) and lays out the traversal records there — what main.cu:524-557
 * (buildBVH + uploads) becomes when the geometry is large. The scene renders exactly like one made by
 * pt_scene_create from the host-built tree. */
pt_scene* pt_scene_create_from_mesh(const pt_scene_desc* desc, int max_leaf_size, pt_bvh_build_stats* stats);
/* Test hook: copy the packed traversal records back (what: 0 nodes 64 B, 1 triangles 48 B, 2 attributes 80 B);
 * returns the record count. */
int pt_debug_packed(pt_scene* scene, int what, void* dst, size_t capacity_bytes);
int novum_bvh_build_host(const pt_float4* positions, int n_positions, const pt_triangle* triangles, int n_triangles,
                         int max_leaf_size, pt_bvh_node* nodes_out, int nodes_capacity,
                         int32_t* indices_out, pt_bvh_build_stats* stats);

/* ---- novum_*: the kept host side (scene loader / initRender) --------------------------- */
typedef struct novum_scene novum_scene;    /* host arrays + RenderConfig + Camera */

/* loadConfig (objects.cuh:844-943) + material table (main.cu:397-467) + readObjSimple per mesh
 * (main.cu:474-482, 936-1068) + computeInfoForBVH/buildBVH (main.cu:524-530) + camera
 * (main.cu:268-273). Mesh paths are resolved against base_dir (NULL = directory of the config). */
novum_scene* novum_scene_load(const char* config_path, const char* base_dir, int render_number);
/* Same, choosing who runs buildBVH: the host (as the reference does) or pt_bvh_build_device (f-4).
 * Both give the same arrays. */
#define NOVUM_BVH_HOST 0
#define NOVUM_BVH_DEVICE 1
novum_scene* novum_scene_load_ex(const char* config_path, const char* base_dir, int render_number, int bvh_builder);
void novum_scene_free(novum_scene* s);
/* info[16]: width,height,spp,maxDepth,integrator,leafSize,nTris,nLights,nNodes,nPoints,nNormals,
 * nUvs,nMats,largestLeaf,backupCount,treeDepth */
void novum_scene_info(const novum_scene* s, int32_t* info16);
void novum_scene_desc(const novum_scene* s, pt_scene_desc* out);     /* pointers stay owned by s */
void novum_scene_camera(const novum_scene* s, pt_camera* out);
/* Camera::Pinhole / Camera::NotPinhole (objects.cuh:221-264) */
void novum_make_camera(int pinhole, const float* pos3, const float* rot3, float fov, float aperture, float focal_dist, int w, int h, pt_camera* out);
/* main.cu:860-870: colors /= spp; NaN -> (1,0,1); Inf -> (0,1,0). n = pixel count. */
void novum_finalise(float* rgba, int n, int sample_count);
/* initRender (main.cu:235-923) for the two unidirectional integrators: load, upload, launch,
 * read back, finalise into out_rgba (w*h float4, may be NULL) and, if bmp_path != NULL, write the
 * tonemapped 24-bit BMP (imageUtil.cu:69-100, 202-232). Returns 0, or < 0 on error. */
int novum_init_render(const char* config_path, const char* base_dir, int render_number, float* out_rgba, const char* bmp_path);
/* Image::saveImageBMP (imageUtil.cu:69-100): rgba is w*h float4 linear radiance, y = 0 bottom. */
int novum_save_bmp(const char* path, const float* rgba, int w, int h, int post_process);
/* Image::saveImageCSV_MONO(channel) (imageUtil.cu:123-142): one channel, scientific, 3 digits. */
int novum_save_csv_mono(const char* path, const float* rgba, int w, int h, int channel);
/* initRender with the reference's progressive preview (deviceCode.cu:574-604): renders in chunks of
 * chunk_spp samples and, whenever at least interval_seconds have passed since the last preview,
 * writes the running average to preview_bmp (and preview_csv if not NULL), as the reference does with
 * render.bmp / renderCSV.csv every 5 s. Final image as novum_init_render. */
int novum_init_render_progressive(const char* config_path, const char* base_dir, int render_number, float* out_rgba,
                                  const char* bmp_path, const char* preview_bmp, const char* preview_csv,
                                  double interval_seconds, int chunk_spp);

#ifdef __cplusplus
}
#endif
#endif /* PT_API_H */
