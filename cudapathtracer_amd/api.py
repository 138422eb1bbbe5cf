"""ctypes binding of ``libptamd.so`` (C ABI: ``include/pt_api.h``).

Python is plumbing here, not the product: every function below is a thin call into the HIP
library. There is NO CPU fallback — loading fails loudly if the library is missing, and every
render fails loudly without a HIP device.

Mirrors the reference's names where it has them: ``launch_unidirectional`` /
``launch_naive_unidirectional`` (deviceCode.cuh:8-12), ``init_render`` (main.cu:235),
``Camera.Pinhole`` / ``NotPinhole`` (objects.cuh:221-264).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_LIB_PATH selects another BUILD of the same HIP library (kernel A/B variants); it is never a fallback.
LIB_PATH = os.environ.get("PT_LIB_PATH") or os.path.join(_HERE, "csrc", "libptamd.so")

UNIDIRECTIONAL = 0
NAIVE_UNIDIRECTIONAL = 2
SEED = 103033  # deviceCode.cu:552
INFO_KEYS = ("width", "height", "spp", "max_depth", "integrator", "leaf_size", "n_tris", "n_lights", "n_nodes",
             "n_points", "n_normals", "n_uvs", "n_mats", "largest_leaf", "backup_count", "tree_depth")
COUNTER_KEYS = ("rays_closest", "rays_shadow", "node_pops", "box_tests", "tri_tests", "hits", "rng_draws", "iterations")


class _F4(C.Union):  # pt_float4: 16 bytes, 16-byte aligned (the long double member only forces the alignment)
    _fields_ = [("v", C.c_float * 4), ("_align", C.c_longdouble)]


class Camera(C.Structure):
    """pt_camera == the reference's Camera (objects.cuh:199-219), 112 bytes."""

    _fields_ = [("cameraOrigin", _F4), ("w", C.c_int32), ("h", C.c_int32), ("xRot", C.c_float), ("yRot", C.c_float),
                ("zRot", C.c_float), ("aperture", C.c_float), ("focalDist", C.c_float), ("fovScale", C.c_float),
                ("antiAliasJitterDist", C.c_float), ("_pad", C.c_float * 3), ("forward", _F4), ("right", _F4), ("up", _F4)]

    @staticmethod
    def Pinhole(pos, w, h, rot=(0.0, 0.0, 0.0), fov=60.0):
        return make_camera(True, pos, rot, fov, w, h)

    @staticmethod
    def NotPinhole(pos, w, h, rot, fov, aperture, focal_dist):
        return make_camera(False, pos, rot, fov, w, h, aperture, focal_dist)

    def tobytes(self):
        return bytes(memoryview(self))

    @staticmethod
    def frombytes(b):
        return Camera.from_buffer_copy(bytes(b))


assert C.sizeof(Camera) == 112 and C.alignment(Camera) == 16


class SceneDesc(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("n_positions", C.c_int32), ("normals", C.c_void_p), ("n_normals", C.c_int32),
                ("uvs", C.c_void_p), ("n_uvs", C.c_int32), ("triangles", C.c_void_p), ("n_triangles", C.c_int32),
                ("lights", C.c_void_p), ("n_lights", C.c_int32), ("bvh", C.c_void_p), ("n_nodes", C.c_int32),
                ("bvh_indices", C.c_void_p), ("materials", C.c_void_p), ("n_materials", C.c_int32),
                ("textures", C.c_void_p), ("n_texels", C.c_int32)]


class TileRange(C.Structure):
    _fields_ = [("first", C.c_int32), ("stride", C.c_int32), ("count", C.c_int32)]


PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p)      # pt_progress_fn


class PtError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libptamd.so; raise if it was not built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtError("%s is missing: build it with `make -C cudapathtracer_amd/csrc` (or __graft_entry__.build()); "
                      "this package has no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: torch ships its own libamdhip64 / libhsa-runtime64, and whichever of
    # two copies initialises second finds "no ROCm-capable device". With torch loaded first, libptamd's
    # DT_NEEDED libamdhip64.so.7 resolves to the copy already in the process. (A C/C++ host that does
    # not use torch links the system ROCm only and has no such issue.)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, u64, f32 = C.c_void_p, C.c_int, C.c_uint64, C.c_float
    L.pt_api_version.restype = i32
    L.pt_last_error.restype = C.c_char_p
    L.pt_device_count.restype = i32
    L.pt_scene_create.restype = vp; L.pt_scene_create.argtypes = [C.POINTER(SceneDesc)]
    L.pt_scene_create_from_mesh.restype = vp; L.pt_scene_create_from_mesh.argtypes = [C.POINTER(SceneDesc), i32, vp]
    L.pt_debug_packed.argtypes = [vp, i32, vp, C.c_size_t]
    L.pt_scene_destroy.argtypes = [vp]
    L.pt_render.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, i32, i32, u64, C.POINTER(TileRange), vp]
    L.pt_render_counted.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, i32, i32, u64, C.POINTER(TileRange), vp, vp]
    L.pt_render_tiles_device.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, i32, i32, u64, C.POINTER(TileRange), vp, i32, vp]
    L.pt_untile_device.argtypes = [i32, i32, C.POINTER(TileRange), vp, vp, vp]
    L.pt_tile_device.argtypes = [i32, i32, C.POINTER(TileRange), vp, vp, vp]
    L.pt_launch_unidirectional.argtypes = [i32, Camera, vp, i32, i32, i32, i32, vp]
    L.pt_launch_naive_unidirectional.argtypes = [i32, Camera, vp, i32, i32, i32, i32, vp]
    L.pt_set_variant.argtypes = [vp, i32]
    L.pt_has_experimental.restype = i32
    L.pt_get_counters.argtypes = [vp, vp]
    L.pt_reset_counters.argtypes = [vp]
    L.pt_last_kernel_ms.restype = f32; L.pt_last_kernel_ms.argtypes = [vp]
    L.pt_scene_flags.argtypes = [vp]
    L.pt_last_tile_handovers.argtypes = [vp]
    L.pt_queue_stalls.argtypes = [vp]
    L.pt_debug_queue_header.argtypes = [vp, C.POINTER(C.c_int)]
    L.pt_set_culling.argtypes = [vp, i32]
    L.pt_set_option.argtypes = [vp, C.c_char_p, i32]
    L.pt_get_option.argtypes = [vp, C.c_char_p, vp]
    L.pt_debug_stamps.argtypes = [vp, vp]
    L.pt_rank_tiles.argtypes = [i32, i32, i32, i32, C.POINTER(TileRange)]
    L.pt_multi_create.restype = vp; L.pt_multi_create.argtypes = [C.POINTER(SceneDesc), i32, vp]
    L.pt_multi_destroy.argtypes = [vp]
    L.pt_multi_set_option.argtypes = [vp, C.c_char_p, i32]
    L.pt_multi_set_variant.argtypes = [vp, i32]
    L.pt_multi_render.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, i32, i32, u64, vp, vp]
    L.pt_render_multi.argtypes = [C.POINTER(SceneDesc), i32, vp, C.POINTER(Camera), i32, i32, i32, i32, i32, i32, u64, vp, vp]
    L.pt_probe_rng.argtypes = [u64, i32, vp, i32, vp, vp, vp]
    L.pt_probe_math.argtypes = [i32, vp, vp, vp, vp, vp, vp]
    L.pt_probe_rcp_exhaustive.argtypes = [vp, vp]
    L.pt_probe_camera_rays.argtypes = [C.POINTER(Camera), u64, i32, vp, vp]
    L.pt_probe_trace_closest.argtypes = [vp, i32, vp, vp, vp, vp]
    L.pt_probe_trace_shadow.argtypes = [vp, i32, vp, vp, vp, vp]
    L.pt_probe_bsdf_sample.argtypes = [vp, i32, vp, vp, vp, f32, f32, u64, vp, vp]
    L.pt_probe_bsdf_eval.argtypes = [vp, i32, vp, vp, vp, f32, f32, vp]
    L.novum_scene_load.restype = vp; L.novum_scene_load.argtypes = [C.c_char_p, C.c_char_p, i32]
    L.novum_scene_load_ex.restype = vp; L.novum_scene_load_ex.argtypes = [C.c_char_p, C.c_char_p, i32, i32]
    L.pt_bvh_build_device.argtypes = [vp, i32, vp, i32, i32, i32, vp, i32, vp, vp]
    L.novum_bvh_build_host.argtypes = [vp, i32, vp, i32, i32, vp, i32, vp, vp]
    L.novum_scene_free.argtypes = [vp]
    L.novum_scene_info.argtypes = [vp, vp]
    L.novum_scene_desc.argtypes = [vp, C.POINTER(SceneDesc)]
    L.novum_scene_camera.argtypes = [vp, C.POINTER(Camera)]
    L.novum_make_camera.argtypes = [i32, vp, vp, f32, f32, f32, i32, i32, C.POINTER(Camera)]
    L.novum_finalise.argtypes = [vp, i32, i32]
    L.novum_init_render.argtypes = [C.c_char_p, C.c_char_p, i32, vp, C.c_char_p]
    L.novum_init_render_progressive.argtypes = [C.c_char_p, C.c_char_p, i32, vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, i32]
    L.novum_save_csv_mono.argtypes = [C.c_char_p, vp, i32, i32, i32]
    L.pt_launch_progressive.argtypes = [i32, i32, Camera, vp, i32, i32, i32, i32, vp, i32, PROGRESS_FN, vp]
    L.novum_save_bmp.argtypes = [C.c_char_p, vp, i32, i32, i32]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _check(rc, what):
    if rc != 0:
        raise PtError("%s failed (%d): %s" % (what, rc, lib().pt_last_error().decode(errors="replace")))


def has_experimental():
    """True if libptamd.so was built with EXPERIMENTAL=1 (the A/B variants of DESIGN.md §6: options wide, compact, spec,
    defer_shadow, xcd_bands, refill = 2). The default library refuses those options with -3."""
    return bool(lib().pt_has_experimental())


def device_count():
    return lib().pt_device_count()


def make_camera(pinhole, pos, rot, fov, w, h, aperture=0.0, focal_dist=0.0):
    cam = Camera()
    pos = np.ascontiguousarray(pos, np.float32); rot = np.ascontiguousarray(rot, np.float32)
    lib().novum_make_camera(int(pinhole), _p(pos), _p(rot), fov, aperture, focal_dist, w, h, C.byref(cam))
    return cam


def finalise(rgba_sum, spp):
    """main.cu:860-870: divide by spp, NaN -> (1,0,1), Inf -> (0,1,0)."""
    out = np.ascontiguousarray(rgba_sum, np.float32).copy()
    lib().novum_finalise(_p(out), out.size // 4, spp)
    return out


def save_bmp(path, rgba, post_process=True):
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[:2]
    if lib().novum_save_bmp(path.encode(), _p(rgba), w, h, int(post_process)) != 0:
        raise PtError("could not write " + path)


def save_csv_mono(path, rgba, channel=0):
    """Image::saveImageCSV_MONO(channel) (imageUtil.cu:123-142)."""
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[:2]
    if lib().novum_save_csv_mono(path.encode(), _p(rgba), w, h, int(channel)) != 0:
        raise PtError("could not write " + path)


def init_render(config_path, render_number=0, base_dir=None, bmp_path=None, preview_bmp=None, preview_csv=None,
                interval_seconds=5.0, chunk_spp=0):
    """initRender (main.cu:235-923) for the unidirectional integrators; returns finalised [h,w,4].
    With chunk_spp > 0 and a preview path it also writes the reference's progressive preview
    (render.bmp / renderCSV.csv every interval_seconds, deviceCode.cu:574-604)."""
    hs = HostScene(config_path, base_dir, render_number)
    w, h = hs.info["width"], hs.info["height"]
    hs.close()
    out = np.zeros((h, w, 4), np.float32)
    enc = lambda s: s.encode() if s else None
    rc = lib().novum_init_render_progressive(config_path.encode(), enc(base_dir), render_number, _p(out), enc(bmp_path),
                                             enc(preview_bmp), enc(preview_csv), float(interval_seconds), int(chunk_spp))
    _check(rc, "novum_init_render")
    return out


BUILD_STATS = np.dtype([("n_nodes", "i4"), ("largest_leaf", "i4"), ("backups", "i4"), ("depth", "i4"), ("sort_fallbacks", "i4"),
                        ("levels", "i4"), ("device_ms", "f4"), ("total_ms", "f4")])


def build_bvh(points, mesh, max_leaf_size, where="device"):
    """buildBVH (main.cu:20-233) on raw arrays in the reference's layouts: `points` float4[n] bytes,
    `mesh` Triangle (80 B)[n] bytes. where="device": pt_bvh_build_device (reference-tree
    mode); where="host": the kept host builder. Returns (nodes uint8[48*n_nodes], indices int32[n], stats dict)."""
    pts = np.ascontiguousarray(points).view(np.uint8).reshape(-1)
    m = np.ascontiguousarray(mesh).view(np.uint8).reshape(-1)
    n = m.size // 80
    cap = max(2 * n - 1, 1)
    nodes = np.zeros(cap * 48, np.uint8)
    idx = np.zeros(max(n, 1), np.int32)
    st = np.zeros(1, BUILD_STATS)
    if where == "device":
        k = lib().pt_bvh_build_device(_p(pts), pts.size // 16, _p(m), n, int(max_leaf_size), 0, _p(nodes), cap, _p(idx), _p(st))
        if k <= 0:
            raise PtError("pt_bvh_build_device failed (%d): %s" % (k, lib().pt_last_error().decode(errors="replace")))
    else:
        k = lib().novum_bvh_build_host(_p(pts), pts.size // 16, _p(m), n, int(max_leaf_size), _p(nodes), cap, _p(idx), _p(st))
        if k <= 0:
            raise PtError("novum_bvh_build_host failed (%d)" % k)
    return nodes[:k * 48].copy(), idx[:n], {f: st[0][f].item() for f in BUILD_STATS.names}


class HostScene:
    """What the kept scene loader produces (novum_scene_load): host arrays in the reference's data model."""

    _ARRAYS = {"points": ("positions", "n_positions", 16), "normals": ("normals", "n_normals", 16), "uvs": ("uvs", "n_uvs", 8),
               "mesh": ("triangles", "n_triangles", 80), "lights": ("lights", "n_lights", 80), "bvh": ("bvh", "n_nodes", 48),
               "indices": ("bvh_indices", "n_triangles", 4), "materials": ("materials", "n_materials", 176),
               "textures": ("textures", "n_texels", 16)}

    def __init__(self, config_path, base_dir=None, render_number=0, bvh_builder="host"):
        """bvh_builder: "host" (the reference's buildBVH on the CPU) or "device" (pt_bvh_build_device; same arrays)."""
        self.h = lib().novum_scene_load_ex(config_path.encode(), base_dir.encode() if base_dir else None, render_number,
                                           {"host": 0, "device": 1}[bvh_builder])
        if not self.h:
            raise PtError("novum_scene_load failed for %s: %s" % (config_path, lib().pt_last_error().decode(errors="replace")))
        info = np.zeros(16, np.int32)
        lib().novum_scene_info(self.h, _p(info))
        self.info = dict(zip(INFO_KEYS, (int(v) for v in info)))
        self.desc = SceneDesc()
        lib().novum_scene_desc(self.h, C.byref(self.desc))

    def camera(self):
        cam = Camera()
        lib().novum_scene_camera(self.h, C.byref(cam))
        return cam

    def array(self, what):
        field, count, size = self._ARRAYS[what]
        n = getattr(self.desc, count) * size
        ptr = getattr(self.desc, field)
        if not ptr or n == 0:
            return np.zeros(0, np.uint8)
        return np.frombuffer((C.c_uint8 * n).from_address(ptr), np.uint8).copy()

    def close(self):
        if self.h:
            lib().novum_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """Device-resident, re-packed scene (pt_scene). Created on the CURRENT HIP device."""

    def __init__(self, host: HostScene | None = None, desc: SceneDesc | None = None, options: dict | None = None):
        d = host.desc if host is not None else desc
        self.h = lib().pt_scene_create(C.byref(d))
        if not self.h:
            raise PtError("pt_scene_create failed: " + lib().pt_last_error().decode(errors="replace"))
        self.set_options(options)

    def set_option(self, name, value):
        """pt_set_option: kernel-selection / scheduling options by name (include/pt_api.h lists them)."""
        _check(lib().pt_set_option(self.h, name.encode(), int(value)), "pt_set_option(%s)" % name)
        return self

    def set_options(self, options):
        for k, v in (options or {}).items():
            self.set_option(k, v)
        return self

    def get_option(self, name):
        out = np.zeros(1, np.int32)
        _check(lib().pt_get_option(self.h, name.encode(), _p(out)), "pt_get_option(%s)" % name)
        return int(out[0])

    @staticmethod
    def from_mesh(host: "HostScene", max_leaf_size=None, options=None):
        """pt_scene_create_from_mesh: BVH build (reference tree) and re-layout on the device from the host scene's
        geometry; its host-built tree is not used. Returns the scene; `.build_stats` has the builder's numbers."""
        st = np.zeros(1, BUILD_STATS)
        leaf = host.info["leaf_size"] if max_leaf_size is None else int(max_leaf_size)
        h = lib().pt_scene_create_from_mesh(C.byref(host.desc), leaf, _p(st))
        if not h:
            raise PtError("pt_scene_create_from_mesh failed: " + lib().pt_last_error().decode(errors="replace"))
        sc = Scene.__new__(Scene)
        sc.h = h; sc._keep = host
        sc.set_options(options)
        sc.build_stats = {f: st[0][f].item() for f in BUILD_STATS.names}
        return sc

    def packed(self, what):
        """Test hook (pt_debug_packed): the traversal records as uint8 [count, record size]; what = nodes / tris / attrs."""
        code, rec = {"nodes": (0, 64), "tris": (1, 48), "attrs": (2, 80)}[what]
        n = lib().pt_debug_packed(self.h, code, None, 0)
        if n < 0:
            raise PtError("pt_debug_packed failed")
        buf = np.zeros((max(n, 0), rec), np.uint8)
        if n:
            lib().pt_debug_packed(self.h, code, _p(buf), buf.nbytes)
        return buf

    @staticmethod
    def from_arrays(arrays, options=None):
        """pt_scene_create straight from arrays in the reference's layouts (dict of buffers: points,
        normals, uvs, mesh, lights, bvh, indices, materials[, textures])."""
        a = {k: np.ascontiguousarray(v).view(np.uint8) for k, v in arrays.items()}
        d = SceneDesc()
        d.positions, d.n_positions = a["points"].ctypes.data, a["points"].size // 16
        d.normals, d.n_normals = a["normals"].ctypes.data, a["normals"].size // 16
        d.uvs, d.n_uvs = a["uvs"].ctypes.data, a["uvs"].size // 8
        d.triangles, d.n_triangles = a["mesh"].ctypes.data, a["mesh"].size // 80
        d.lights, d.n_lights = (a["lights"].ctypes.data if a["lights"].size else None), a["lights"].size // 80
        d.bvh, d.n_nodes = a["bvh"].ctypes.data, a["bvh"].size // 48
        d.bvh_indices = a["indices"].ctypes.data
        d.materials, d.n_materials = a["materials"].ctypes.data, a["materials"].size // 176
        t = a.get("textures")
        d.textures, d.n_texels = (t.ctypes.data if t is not None and t.size else None), (t.size // 16 if t is not None else 0)
        return Scene(desc=d, options=options)

    @staticmethod
    def from_config(config_path, base_dir=None, render_number=0):
        hs = HostScene(config_path, base_dir, render_number)
        return Scene(hs), hs

    def close(self):
        if self.h:
            lib().pt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- launchers ------------------------------------------------------------------------------
    def render(self, camera, w, h, spp, max_depth, integrator=UNIDIRECTIONAL, use_mis=True, seed=SEED, tiles=None,
               counters=False, out=None):
        """Host-buffer launcher: returns (sum-of-samples [h,w,4] float32, per-pixel counters [h,w,8] or None)."""
        col = np.zeros((h, w, 4), np.float32) if out is None else out
        cnt = np.zeros((h, w, 8), np.uint32) if counters else None
        tr = C.byref(tiles) if tiles is not None else None
        if counters:
            rc = lib().pt_render_counted(self.h, C.byref(camera), w, h, spp, max_depth, integrator, int(use_mis), seed, tr, _p(col), _p(cnt))
        else:                                               # the timed kernels (time slices, REFILL / FLAT instantiations)
            rc = lib().pt_render(self.h, C.byref(camera), w, h, spp, max_depth, integrator, int(use_mis), seed, tr, _p(col))
        _check(rc, "pt_render")
        return col, cnt

    def render_tiles_device(self, camera, w, h, spp, max_depth, d_tile_ptr, integrator=UNIDIRECTIONAL, use_mis=True,
                            seed=SEED, tiles=None, count_work=False, stream=0):
        tr = C.byref(tiles) if tiles is not None else None
        rc = lib().pt_render_tiles_device(self.h, C.byref(camera), w, h, spp, max_depth, integrator, int(use_mis), seed, tr,
                                          d_tile_ptr, int(count_work), stream or None)
        _check(rc, "pt_render_tiles_device")

    def launch_unidirectional(self, max_depth, camera, num_sample, use_mis, w, h, d_colors_ptr):
        _check(lib().pt_launch_unidirectional(max_depth, camera, self.h, num_sample, int(use_mis), w, h, d_colors_ptr), "pt_launch_unidirectional")

    def launch_progressive(self, integrator, max_depth, camera, num_sample, use_mis, w, h, d_colors_ptr, chunk_spp, progress=None):
        """pt_launch_progressive: progress(samples_done) -> truthy to stop early."""
        cb = PROGRESS_FN((lambda done, _u: int(bool(progress(done)))) if progress else (lambda done, _u: 0))
        _check(lib().pt_launch_progressive(integrator, max_depth, camera, self.h, num_sample, int(use_mis), w, h, d_colors_ptr, chunk_spp, cb, None),
               "pt_launch_progressive")

    def launch_naive_unidirectional(self, max_depth, camera, num_sample, use_mis, w, h, d_colors_ptr):
        _check(lib().pt_launch_naive_unidirectional(max_depth, camera, self.h, num_sample, int(use_mis), w, h, d_colors_ptr), "pt_launch_naive_unidirectional")

    def set_variant(self, variant):
        """0 / "megakernel" (default) or 1 / "wavefront" (stream-compacted A/B variant; same results)."""
        v = {"megakernel": 0, "wavefront": 1}.get(variant, variant)
        _check(lib().pt_set_variant(self.h, int(v)), "pt_set_variant")
        return self

    def counters(self):
        out = np.zeros(8, np.uint64)
        _check(lib().pt_get_counters(self.h, _p(out)), "pt_get_counters")
        return dict(zip(COUNTER_KEYS, (int(v) for v in out)))

    def reset_counters(self):
        _check(lib().pt_reset_counters(self.h), "pt_reset_counters")

    def debug_stamps(self):
        out = np.zeros(8, np.uint64)
        _check(lib().pt_debug_stamps(self.h, _p(out)), "pt_debug_stamps")
        return dict(zip(("regen", "closest", "bounce_logic", "wave_lifetimes", "not_earliest_start", "latest_end", "shadow_in_bounce", "slot7"), (int(v) for v in out)))

    def global_node_fetches(self):
        """Counting launches since reset_counters: internal-node fetches that missed the LDS scene cache (normal builds)."""
        out = np.zeros(8, np.uint64)
        _check(lib().pt_debug_stamps(self.h, _p(out)), "pt_debug_stamps")
        return int(out[0])

    def debug_lane_util(self):
        """-DPT_UTIL builds, after a counting render: lanes carried per trip through the traversal loops."""
        out = np.zeros(8, np.uint64)
        _check(lib().pt_debug_stamps(self.h, _p(out)), "pt_debug_stamps")
        v = [int(x) for x in out]
        names = ("closest_nodes", "closest_tris", "shadow_nodes", "shadow_tris")
        return {n: {"wave_trips": v[2 * k], "lane_trips": v[2 * k + 1], "lanes_per_trip": (v[2 * k + 1] / v[2 * k]) if v[2 * k] else 0.0}
                for k, n in enumerate(names)}

    def set_culling(self, on=True):
        """pt_set_culling: opt-in box culling (not the reference's visiting set; see pt_api.h)."""
        _check(lib().pt_set_culling(self.h, int(bool(on))), "pt_set_culling")
        return self

    def flags(self):
        f = lib().pt_scene_flags(self.h)
        return {"onchip": bool(f & 1), "persistent": bool(f & 2), "time_slices": bool(f & 4), "hbm_kernel": bool(f & 8), "culling": bool(f & 16), "refill": bool(f & 32), "flat": bool(f & 64), "simple": bool(f & 128), "flat_pair": bool(f & 256), "leaf_table": bool(f & 512), "lean": bool(f & 1024)}

    def tile_handovers(self):
        """pt_last_tile_handovers: tiles that changed hands between waves in the last (completed) megakernel launch."""
        n = lib().pt_last_tile_handovers(self.h)
        _check(min(n, 0), "pt_last_tile_handovers")
        return n

    def queue_header(self):
        """pt_debug_queue_header: the tile queue's 16 header words after the last queued launch (None: that launch used no queue)."""
        out = (C.c_int * 16)()
        rc = lib().pt_debug_queue_header(self.h, out)
        if rc < 0:
            _check(rc, "pt_debug_queue_header")
        return list(out) if rc == 1 else None

    def queue_stalls(self):
        """pt_queue_stalls: launches whose queue waiters gave up although the frame was complete (not an error)."""
        return int(lib().pt_queue_stalls(self.h))

    def last_kernel_ms(self):
        """Device time of the last launch; raises if that launch did not finish its frame (tile-queue timeout).
        Callers of render_tiles_device (asynchronous) learn about an incomplete frame here."""
        ms = float(lib().pt_last_kernel_ms(self.h))
        if ms < 0.0:
            raise PtError("pt_last_kernel_ms: " + (lib().pt_last_error().decode(errors="replace") or "the last launch failed"))
        return ms

    # -- probes ---------------------------------------------------------------------------------
    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = len(rays)
        oi = np.zeros((n, 4), np.int32); of = np.zeros((n, 12), np.float32); cnt = np.zeros(8, np.uint64)
        _check(lib().pt_probe_trace_closest(self.h, n, _p(rays), _p(oi), _p(of), _p(cnt)), "pt_probe_trace_closest")
        return oi, of, dict(zip(COUNTER_KEYS, (int(v) for v in cnt)))

    def trace_shadow(self, rays, max_t):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        max_t = np.ascontiguousarray(max_t, np.float32)
        n = len(rays)
        of = np.zeros((n, 3), np.float32); cnt = np.zeros(8, np.uint64)
        _check(lib().pt_probe_trace_shadow(self.h, n, _p(rays), _p(max_t), _p(of), _p(cnt)), "pt_probe_trace_shadow")
        return of, dict(zip(COUNTER_KEYS, (int(v) for v in cnt)))

    def bsdf_sample(self, material, wi, backface, subseq, eta_i=1.0, eta_t=1.0, seed=SEED):
        material = np.ascontiguousarray(material, np.int32); wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3)
        backface = np.ascontiguousarray(backface, np.int32); subseq = np.ascontiguousarray(subseq, np.uint32)
        n = len(material)
        out = np.zeros((n, 8), np.float32)
        _check(lib().pt_probe_bsdf_sample(self.h, n, _p(material), _p(wi), _p(backface), eta_i, eta_t, seed, _p(subseq), _p(out)), "pt_probe_bsdf_sample")
        return out

    def bsdf_eval(self, material, wi, wo, eta_i=1.0, eta_t=1.0):
        material = np.ascontiguousarray(material, np.int32)
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3); wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        n = len(material)
        out = np.zeros((n, 4), np.float32)
        _check(lib().pt_probe_bsdf_eval(self.h, n, _p(material), _p(wi), _p(wo), eta_i, eta_t, _p(out)), "pt_probe_bsdf_eval")
        return out


MULTI_STATS = np.dtype([("n_devices", "i4"), ("gather", "i4"), ("kernel_ms", "f4", (16,)), ("render_ms", "f4"), ("gather_ms", "f4"), ("total_ms", "f4")])


class MultiScene:
    """pt_multi: one replica of the scene per HIP device, frame sharded by interleaved 8x8 tiles, one gather to device 0
    (include/pt_api.h, "multi-GPU"). device_ids=None means devices 0 .. n_devices-1."""

    def __init__(self, host: "HostScene", n_devices, device_ids=None, options=None):
        ids = np.ascontiguousarray(device_ids, np.int32) if device_ids is not None else None
        self._keep = host
        self.h = lib().pt_multi_create(C.byref(host.desc), int(n_devices), _p(ids))
        if not self.h:
            raise PtError("pt_multi_create failed: " + lib().pt_last_error().decode(errors="replace"))
        for k, v in (options or {}).items():
            self.set_option(k, v)
        self.stats = None

    def set_option(self, name, value):
        _check(lib().pt_multi_set_option(self.h, name.encode(), int(value)), "pt_multi_set_option(%s)" % name)
        return self

    def set_variant(self, variant):
        _check(lib().pt_multi_set_variant(self.h, int({"megakernel": 0, "wavefront": 1}.get(variant, variant))), "pt_multi_set_variant")
        return self

    def render(self, camera, w, h, spp, max_depth, integrator=UNIDIRECTIONAL, use_mis=True, seed=SEED, out=None):
        col = np.zeros((h, w, 4), np.float32) if out is None else out
        st = np.zeros(1, MULTI_STATS)
        _check(lib().pt_multi_render(self.h, C.byref(camera), w, h, spp, max_depth, integrator, int(use_mis), seed, _p(col), _p(st)), "pt_multi_render")
        self.stats = {"n_devices": int(st[0]["n_devices"]), "gather": {0: "none", 1: "rccl", 2: "peer_copy"}[int(st[0]["gather"])],
                      "kernel_ms": [float(v) for v in st[0]["kernel_ms"][:int(st[0]["n_devices"])]],
                      "render_ms": float(st[0]["render_ms"]), "gather_ms": float(st[0]["gather_ms"]), "total_ms": float(st[0]["total_ms"])}
        return col

    def close(self):
        if self.h:
            lib().pt_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_multi(host, n_devices, camera, w, h, spp, max_depth, integrator=UNIDIRECTIONAL, use_mis=True, seed=SEED, device_ids=None):
    """pt_render_multi: the one-shot form."""
    ids = np.ascontiguousarray(device_ids, np.int32) if device_ids is not None else None
    col = np.zeros((h, w, 4), np.float32)
    _check(lib().pt_render_multi(C.byref(host.desc), int(n_devices), _p(ids), C.byref(camera), w, h, spp, max_depth, integrator, int(use_mis), seed, _p(col), None),
           "pt_render_multi")
    return col


def parse_options(pairs):
    """["flat=0", "waves_hbm=2"] -> {"flat": 0, "waves_hbm": 2} for the --opt flag of bench.py and tools/."""
    out = {}
    for p in pairs or []:
        for item in p.split(","):
            if item:
                k, _, v = item.partition("=")
                out[k.strip()] = int(v)
    return out


def untile_device(w, h, d_tiles_ptr, d_colors_ptr, tiles=None, stream=0):
    _check(lib().pt_untile_device(w, h, C.byref(tiles) if tiles is not None else None, d_tiles_ptr, d_colors_ptr, stream or None), "pt_untile_device")


def tile_device(w, h, d_colors_ptr, d_tiles_ptr, tiles=None, stream=0):
    _check(lib().pt_tile_device(w, h, C.byref(tiles) if tiles is not None else None, d_colors_ptr, d_tiles_ptr, stream or None), "pt_tile_device")


def n_tiles(w, h):
    return ((w + 7) // 8) * ((h + 7) // 8)


def rank_tiles(w, h, rank, world):
    """Interleaved tile ownership: rank r renders tiles {t : t mod world == r} (pt_rank_tiles)."""
    tr = TileRange()
    lib().pt_rank_tiles(w, h, rank, world, C.byref(tr))
    return tr


def probe_rng(subsequences, n_draws, seed=SEED):
    sub = np.ascontiguousarray(subsequences, np.uint32)
    n = len(sub)
    st = np.zeros((n, 6), np.uint32); u = np.zeros((n, max(n_draws, 1)), np.uint32); f = np.zeros((n, max(n_draws, 1)), np.float32)
    _check(lib().pt_probe_rng(seed, n, _p(sub), n_draws, _p(st), _p(u), _p(f)), "pt_probe_rng")
    return st, u[:, :n_draws], f[:, :n_draws]


def probe_math(x):
    x = np.ascontiguousarray(x, np.float32)
    outs = [np.zeros_like(x) for _ in range(5)]
    _check(lib().pt_probe_math(x.size, _p(x), *[_p(o) for o in outs]), "pt_probe_math")
    return dict(zip(("sin", "cos", "exp", "rsqrt", "pow5"), outs))


def probe_rcp_exhaustive():
    """pt_probe_rcp_exhaustive: {mismatches, in_fast_range, bare_sequence_wrong_outside_range, first_bad} over all 2^32 inputs."""
    out = np.zeros(3, np.uint64)
    first = np.zeros(1, np.uint32)
    _check(lib().pt_probe_rcp_exhaustive(_p(out), _p(first)), "pt_probe_rcp_exhaustive")
    return {"mismatches": int(out[0]), "in_fast_range": int(out[1]), "bare_wrong_outside": int(out[2]), "first_bad": int(first[0])}


def probe_camera_rays(camera, xy, seed=SEED):
    xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
    out = np.zeros((len(xy), 6), np.float32)
    _check(lib().pt_probe_camera_rays(C.byref(camera), seed, len(xy), _p(xy), _p(out)), "pt_probe_camera_rays")
    return out
