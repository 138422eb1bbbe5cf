// pt_multi.hip — the multi-GPU form of the launcher boundary (include/pt_api.h "multi-GPU").
//
// The reference is single-GPU (one launch_unidirectional per frame, main.cu:565). Every pixel's estimate depends only on
// (scene, camera, GLOBAL pixel index, seed) — the XORWOW stream is keyed by y*w+x (deviceCode.cu:59-60) — so the frame
// shards by screen tile with no data-path collective: device r of N renders the 8x8 tiles {t : t mod N == r}
// (interleaved: per-tile cost varies with geometry and path length) of a replicated scene, and ONE gather brings the
// tile buffers to device 0, which de-interleaves them into scan-line order. The image is bit-identical for any N.
//
// Built on the public C ABI only (pt_scene_create, pt_render_tiles_device, pt_untile_device): one host thread per device,
// as a C++ host of the reference would do it. The threads share nothing writable: each cuts its own tiles out of the
// caller's frame into its own pinned staging buffer (the frame crosses PCIe once in total, not once per device), owns its
// scene replica, stream, events and error string, and only device 0's thread touches the gather buffer and the output. The gather is RCCL point-to-point (grouped
// ncclSend / ncclRecv = ncclGather; xGMI is point-to-point, 33 MB at 1080p, one hop per peer, no ring) with
// hipMemcpyPeerAsync as the second transport. RCCL is bound at run time (dlopen of librccl.so.1: inside a PyTorch
// process that resolves to the copy already loaded, in a plain C++ host to /opt/rocm/lib), so that single-GPU users of
// libptamd.so do not map a 570 MB library they never call.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pt_api.h"

extern "C" int pt_fail_(int code, const char* msg);

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    bool load() {
        if (handle) return true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);                 // the copy this process already uses, if any
            if (!handle) handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { error = std::string("librccl.so.1 not loadable: ") + (dlerror() ? dlerror() : "?"); return false; }
#define PT_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(handle, #sym)); if (!field) { error = "RCCL symbol " #sym " missing"; return false; }
        PT_SYM(CommInitAll, ncclCommInitAll) PT_SYM(CommDestroy, ncclCommDestroy) PT_SYM(GroupStart, ncclGroupStart)
        PT_SYM(GroupEnd, ncclGroupEnd) PT_SYM(Send, ncclSend) PT_SYM(Recv, ncclRecv) PT_SYM(GetErrorString, ncclGetErrorString)
#undef PT_SYM
        return true;
    }
};
Rccl g_rccl;

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Rank {
    int device = 0;
    pt_scene* scene = nullptr;
    hipStream_t stream = nullptr;
    void* dColors = nullptr; size_t colorsBytes = 0;    // rank 0 only: the scan-line frame the gathered tiles are de-interleaved into
    void* dTiles = nullptr; size_t tilesBytes = 0;      // this rank's tile-major buffer, padded to the common count
    void* hStage = nullptr; size_t stageBytes = 0;      // this rank's OWN pinned staging buffer: its tiles of the caller's frame, cut out on the host
    ncclComm_t comm = nullptr;
    int rc = 0; std::string err;
    float kernelMs = 0.0f;
};

}  // namespace

struct pt_multi {
    // Ranks that share a device (rehearsals on a box with fewer GPUs than ranks; never the case on a real node). The
    // megakernel is persistent — its grid fills the chip and its waves wait on a tile queue — so several of them on one
    // device only oversubscribe it: by default such ranks issue their work to the stream of the FIRST rank on that device
    // and the kernels run in stream order ("same_device" 1). All host threads still run concurrently — every host-side
    // object of the library (scenes, events, staging buffers, error strings) is exercised as on a real node — and no
    // host lock is involved. "same_device" 0 gives every rank its own stream: the kernels then co-reside on the device.
    int sameDeviceOrdered = 1;
    int n = 0;
    std::vector<Rank> ranks;
    void* dGather = nullptr; size_t gatherBytes = 0;    // on device ranks[0].device: [rank][padded tiles][64] float4
    int gather = 0;                                      // 0 auto (RCCL if loadable, else peer copies), 1 RCCL, 2 peer copies
    int selfGather = 0;                                  // n == 1: run the collective anyway (rank 0 sends to itself) — plumbing test
    bool commsReady = false;
    int lastTransport = 0;
};

namespace {

int ensure(void*& p, size_t& have, size_t need) {
    if (need <= have) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; have = 0;
    if (hipMalloc(&p, std::max<size_t>(need, 16)) != hipSuccess) return -2;
    have = need;
    return 0;
}
int ensure_pinned(void*& p, size_t& have, size_t need) {
    if (need <= have) return 0;
    if (p) (void)hipHostFree(p);
    p = nullptr; have = 0;
    if (hipHostMalloc(&p, std::max<size_t>(need, 16), hipHostMallocDefault) != hipSuccess) return -2;
    have = need;
    return 0;
}

// The tiles of `tr` cut out of the scan-line frame `colors` into tile-major order [local tile][ly * 8 + lx] (what tile_kernel
// does on the device): lanes outside the image and the padding tiles up to `pad` are zero. Each rank reads only ITS pixels of
// the caller's frame — 1 / N of it — and uploads them from its own pinned buffer, so the N host threads share nothing
// writable and the frame crosses PCIe once in total instead of once per device.
void tile_host(const float* colors, int w, int h, const pt_tile_range& tr, int pad, float* tiles) {
    const int tilesX = (w + 7) / 8;
    memset(tiles, 0, (size_t)pad * 64 * 16);
    for (int lt = 0; lt < tr.count; lt++) {
        const int tile = tr.first + lt * tr.stride, x0 = (tile % tilesX) * 8, y0 = (tile / tilesX) * 8;
        const int xs = std::min(8, w - x0), ys = std::min(8, h - y0);
        for (int ly = 0; ly < ys; ly++)
            memcpy(tiles + ((size_t)lt * 64 + (size_t)ly * 8) * 4, colors + ((size_t)(y0 + ly) * w + x0) * 4, (size_t)xs * 16);
    }
}

// The stream a rank's device work goes to: its own, or (ranks sharing a device, "same_device" 1) the first such rank's.
hipStream_t stream_of(const pt_multi* m, int r) {
    if (m->sameDeviceOrdered)
        for (int q = 0; q < r; q++) if (m->ranks[q].device == m->ranks[r].device) return m->ranks[q].stream;
    return m->ranks[r].stream;
}

template <class F>
void on_each_rank(pt_multi* m, F f) {
    std::vector<std::thread> th;
    for (int r = 0; r < m->n; r++)
        th.emplace_back([m, r, &f] {
            Rank& k = m->ranks[r];
            k.rc = 0; k.err.clear();
            if (hipSetDevice(k.device) != hipSuccess) { k.rc = -2; k.err = "hipSetDevice failed"; return; }
            k.rc = f(r, k);
            if (k.rc != 0 && k.err.empty()) k.err = pt_last_error();       // pt_last_error is per thread: carry it out
        });
    for (auto& t : th) t.join();
}

int first_error(pt_multi* m, const char* what) {
    for (int r = 0; r < m->n; r++)
        if (m->ranks[r].rc != 0) {
            char buf[600];
            snprintf(buf, sizeof(buf), "%s: device %d (rank %d of %d): %s", what, m->ranks[r].device, r, m->n, m->ranks[r].err.c_str());
            return pt_fail_(m->ranks[r].rc, buf);
        }
    return 0;
}

int init_comms(pt_multi* m) {
    if (m->commsReady) return 0;
    if (!g_rccl.load()) return pt_fail_(-5, g_rccl.error.c_str());
    std::vector<int> devs(m->n);
    for (int r = 0; r < m->n; r++) devs[r] = m->ranks[r].device;
    std::vector<ncclComm_t> comms(m->n, nullptr);
    ncclResult_t e = g_rccl.CommInitAll(comms.data(), m->n, devs.data());
    if (e != ncclSuccess) {
        char buf[300];
        snprintf(buf, sizeof(buf), "ncclCommInitAll over %d devices failed: %s", m->n, g_rccl.GetErrorString(e));
        return pt_fail_(-5, buf);
    }
    for (int r = 0; r < m->n; r++) m->ranks[r].comm = comms[r];
    m->commsReady = true;
    return 0;
}

}  // namespace

extern "C" {

void pt_rank_tiles(int w, int h, int rank, int world, pt_tile_range* out) {
    if (!out) return;
    const int total = ((w + 7) / 8) * ((h + 7) / 8);
    out->first = rank; out->stride = world;
    out->count = (world > 0 && rank >= 0 && rank < total) ? (total - rank + world - 1) / world : 0;
}

void pt_multi_destroy(pt_multi* m) {
    if (!m) return;
    for (Rank& k : m->ranks) {
        if (hipSetDevice(k.device) != hipSuccess) continue;
        if (k.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(k.comm);
        if (k.scene) pt_scene_destroy(k.scene);
        if (k.dColors) (void)hipFree(k.dColors);
        if (k.dTiles) (void)hipFree(k.dTiles);
        if (k.hStage) (void)hipHostFree(k.hStage);
        if (k.stream) (void)hipStreamDestroy(k.stream);
    }
    if (m->dGather && !m->ranks.empty() && hipSetDevice(m->ranks[0].device) == hipSuccess) (void)hipFree(m->dGather);
    delete m;
}

pt_multi* pt_multi_create(const pt_scene_desc* desc, int n_devices, const int* device_ids) {
    if (!desc) { pt_fail_(-1, "pt_multi_create: null desc"); return nullptr; }
    const int avail = pt_device_count();
    if (avail <= 0) { pt_fail_(-2, "pt_multi_create: no usable HIP device; the path has no CPU fallback"); return nullptr; }
    if (n_devices < 1 || n_devices > PT_MULTI_MAX_DEVICES) { pt_fail_(-1, "pt_multi_create: n_devices out of range"); return nullptr; }
    pt_multi* m = new pt_multi();
    m->n = n_devices;
    m->ranks.resize(n_devices);
    for (int r = 0; r < n_devices; r++) {
        const int d = device_ids ? device_ids[r] : r;
        if (d < 0 || d >= avail) {
            char buf[200];
            snprintf(buf, sizeof(buf), "pt_multi_create: device id %d (rank %d) but %d HIP device(s) are visible", d, r, avail);
            pt_fail_(-1, buf);
            delete m;
            return nullptr;
        }
        m->ranks[r].device = d;
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    // replicate the scene: every rank re-packs and uploads in its own thread (main.cu:469-557 per device)
    on_each_rank(m, [desc](int, Rank& k) -> int {
        k.scene = pt_scene_create(desc);
        if (!k.scene) return -2;
        if (hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) != hipSuccess) { k.err = "hipStreamCreate failed"; return -2; }
        return 0;
    });
    (void)hipSetDevice(prev);
    if (first_error(m, "pt_multi_create") != 0) { pt_multi_destroy(m); return nullptr; }
    // direct peer copies where the fabric allows them (xGMI); without it hipMemcpyPeerAsync stages through the host
    for (int r = 1; r < n_devices; r++) {
        if (m->ranks[r].device == m->ranks[0].device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->ranks[r].device, m->ranks[0].device) == hipSuccess && can && hipSetDevice(m->ranks[r].device) == hipSuccess)
            (void)hipDeviceEnablePeerAccess(m->ranks[0].device, 0);
    }
    (void)hipGetLastError();
    (void)hipSetDevice(prev);
    return m;
}

int pt_multi_set_option(pt_multi* m, const char* name, int value) {
    if (!m || !name) return pt_fail_(-1, "pt_multi_set_option: null argument");
    if (!strcmp(name, "gather")) {
        if (value < 0 || value > 2) return pt_fail_(-1, "pt_multi_set_option: gather is 0 (auto), 1 (RCCL) or 2 (peer copies)");
        m->gather = value;
        return 0;
    }
    if (!strcmp(name, "self_gather")) { m->selfGather = value != 0; return 0; }
    if (!strcmp(name, "same_device")) {
        if (value < 0 || value > 1) return pt_fail_(-1, "pt_multi_set_option: same_device is 0 (own streams: kernels co-reside) or 1 (one stream per device)");
        m->sameDeviceOrdered = value;
        return 0;
    }
    for (Rank& k : m->ranks)                                                   // everything else: per-scene options of every replica
        if (int rc = pt_set_option(k.scene, name, value)) return rc;
    return 0;
}

int pt_multi_set_variant(pt_multi* m, int variant) {
    if (!m) return pt_fail_(-1, "null handle");
    for (Rank& k : m->ranks)
        if (int rc = pt_set_variant(k.scene, variant)) return rc;
    return 0;
}

int pt_multi_render(pt_multi* m, const pt_camera* cam, int w, int h, int spp, int max_depth, int integrator, int use_mis,
                    uint64_t seed, float* out_rgba_sum, pt_multi_stats* stats) {
    if (!m || !cam || !out_rgba_sum) return pt_fail_(-1, "pt_multi_render: null argument");
    if (w <= 0 || h <= 0) return pt_fail_(-1, "pt_multi_render: bad image size");
    const int N = m->n;
    const size_t frameBytes = (size_t)w * h * 16;
    const int total = ((w + 7) / 8) * ((h + 7) / 8);
    const int pad = (total + N - 1) / N;                                   // equal counts for the gather
    const size_t slotBytes = (size_t)pad * 64 * 16;
    const bool collective = N > 1 || m->selfGather;
    int transport = 0;                                                    // 1 RCCL, 2 peer copies
    if (collective) {
        transport = m->gather == 2 ? 2 : 1;
        if (transport == 1) {
            bool dup = false;                                              // RCCL refuses one device twice (rehearsals on a one-GPU box)
            for (int a = 0; a < N; a++) for (int b = a + 1; b < N; b++) dup |= m->ranks[a].device == m->ranks[b].device;
            if (dup || init_comms(m) != 0) {
                if (m->gather == 1) return dup ? pt_fail_(-5, "pt_multi_render: RCCL cannot run two ranks on one device") : -5;
                transport = 2;
            }
        }
    }
    m->lastTransport = transport;
    int prev = 0;
    (void)hipGetDevice(&prev);
    const double t0 = now_ms();

    // ---- phase 1: every rank cuts ITS tiles out of the caller's frame (host, own pinned buffer), uploads and renders them ----
    on_each_rank(m, [&](int r, Rank& k) -> int {
        pt_tile_range tr;
        pt_rank_tiles(w, h, r, N, &tr);
        if (ensure(k.dTiles, k.tilesBytes, slotBytes) || ensure_pinned(k.hStage, k.stageBytes, slotBytes)) { k.err = "allocation of the tile buffers failed"; return -2; }
        if (r == 0 && (ensure(k.dColors, k.colorsBytes, frameBytes) || ensure(m->dGather, m->gatherBytes, slotBytes * (size_t)N))) { k.err = "hipMalloc failed"; return -2; }
        hipStream_t st = stream_of(m, r);
        tile_host(out_rgba_sum, w, h, tr, pad, (float*)k.hStage);          // colors[pixelIdx] += Li: the sum starts from `out`
        if (hipMemcpyAsync(k.dTiles, k.hStage, slotBytes, hipMemcpyHostToDevice, st) != hipSuccess) { k.err = "H2D copy failed"; return -2; }
        if (int rc = pt_render_tiles_device(k.scene, cam, w, h, spp, max_depth, integrator, use_mis, seed, &tr, k.dTiles, 0, st)) return rc;
        if (hipStreamSynchronize(st) != hipSuccess) { k.err = "stream synchronise failed after the render"; return -2; }
        k.kernelMs = pt_last_kernel_ms(k.scene);                            // also reads the tile queue's error word
        return k.kernelMs < 0.0f ? -4 : 0;
    });
    const double t1 = now_ms();
    if (int rc = first_error(m, "pt_multi_render")) { (void)hipSetDevice(prev); return rc; }

    // ---- phase 2: the path's single collective — tile buffers to device 0 ----
    on_each_rank(m, [&](int r, Rank& k) -> int {
        char* gather0 = (char*)m->dGather;
        if (!collective) return 0;
        hipStream_t st = stream_of(m, r);
        if (transport == 1) {
            const size_t count = slotBytes / 4;
            ncclResult_t e = g_rccl.GroupStart();
            if (r == 0) {
                for (int p = (N > 1 ? 1 : 0); p < N && e == ncclSuccess; p++) e = g_rccl.Recv(gather0 + (size_t)p * slotBytes, count, ncclFloat, p, k.comm, st);
                if (N == 1 && e == ncclSuccess) e = g_rccl.Send(k.dTiles, count, ncclFloat, 0, k.comm, st);      // self_gather rehearsal
            } else {
                e = g_rccl.Send(k.dTiles, count, ncclFloat, 0, k.comm, st);
            }
            ncclResult_t e2 = g_rccl.GroupEnd();
            if (e == ncclSuccess) e = e2;
            if (e != ncclSuccess) { k.err = std::string("RCCL gather failed: ") + g_rccl.GetErrorString(e); return -5; }
        } else if (r > 0 || N == 1) {
            if (hipMemcpyPeerAsync(gather0 + (size_t)r * slotBytes, m->ranks[0].device, k.dTiles, k.device, slotBytes, st) != hipSuccess) {
                k.err = "hipMemcpyPeerAsync failed"; return -2;
            }
        }
        if (hipStreamSynchronize(st) != hipSuccess) { k.err = "stream synchronise failed after the gather"; return -2; }
        return 0;
    });
    const double t2 = now_ms();
    if (int rc = first_error(m, "pt_multi_render (gather)")) { (void)hipSetDevice(prev); return rc; }

    // ---- phase 3: device 0 de-interleaves into scan-line order and hands the frame back ----
    int rc = 0;
    {
        Rank& k0 = m->ranks[0];
        if (hipSetDevice(k0.device) != hipSuccess) rc = pt_fail_(-2, "hipSetDevice failed");
        for (int p = 0; p < N && rc == 0; p++) {
            pt_tile_range tr;
            pt_rank_tiles(w, h, p, N, &tr);
            const void* src = (p == 0 && !(N == 1 && collective)) ? k0.dTiles : (const void*)((char*)m->dGather + (size_t)p * slotBytes);
            rc = pt_untile_device(w, h, &tr, src, k0.dColors, k0.stream);
        }
        if (rc == 0 && hipMemcpyAsync(out_rgba_sum, k0.dColors, frameBytes, hipMemcpyDeviceToHost, k0.stream) != hipSuccess) rc = pt_fail_(-2, "D2H copy failed");
        if (rc == 0 && hipStreamSynchronize(k0.stream) != hipSuccess) rc = pt_fail_(-2, "stream synchronise failed after the read-back");
    }
    const double t3 = now_ms();
    (void)hipSetDevice(prev);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->n_devices = N; stats->gather = transport;
        for (int r = 0; r < N && r < PT_MULTI_MAX_DEVICES; r++) stats->kernel_ms[r] = m->ranks[r].kernelMs;
        stats->render_ms = (float)(t1 - t0); stats->gather_ms = (float)(t2 - t1); stats->total_ms = (float)(t3 - t0);
    }
    return rc;
}

int pt_render_multi(const pt_scene_desc* desc, int n_devices, const int* device_ids, const pt_camera* cam, int w, int h, int spp,
                    int max_depth, int integrator, int use_mis, uint64_t seed, float* out_rgba_sum, pt_multi_stats* stats) {
    pt_multi* m = pt_multi_create(desc, n_devices, device_ids);
    if (!m) return -2;
    const int rc = pt_multi_render(m, cam, w, h, spp, max_depth, integrator, use_mis, seed, out_rgba_sum, stats);
    const std::string keep = rc ? pt_last_error() : "";
    pt_multi_destroy(m);
    if (rc) pt_fail_(rc, keep.c_str());
    return rc;
}

}  // extern "C"
