"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE (oracle/README.md).

Import this only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
PARITY UNPINNED (no reference golden vectors exist; see oracle/README.md).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

INFO_KEYS = ("width", "height", "spp", "max_depth", "integrator", "leaf_size", "n_tris", "n_lights", "n_nodes",
             "n_points", "n_normals", "n_uvs", "n_mats", "largest_leaf", "backup_count", "tree_depth")
COUNTER_KEYS = ("rays_closest", "rays_shadow", "node_pops", "box_tests", "tri_tests", "hits", "rng_draws", "iterations")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_scene_load.restype = C.c_void_p
        L.oracle_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.oracle_scene_from_arrays.restype = C.c_void_p
        L.oracle_scene_from_arrays.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.oracle_build_bvh.restype = C.c_int
        L.oracle_build_bvh.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_scene_texels.restype = C.c_int
        L.oracle_scene_texels.argtypes = [C.c_void_p]
        L.oracle_scene_free.argtypes = [C.c_void_p]
        L.oracle_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_scene_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_make_camera.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.oracle_render.restype = C.c_double
        L.oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulonglong,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_finalise.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_save_bmp.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_save_csv_mono.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_tonemap_gamma.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_xorwow_init.argtypes = [C.c_ulonglong, C.c_uint, C.c_void_p]
        L.oracle_xorwow_next.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_xorwow_uniform.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_xorwow_matrix.argtypes = [C.c_int, C.c_void_p]
        for f in (L.oracle_expf, L.oracle_rsqrtf, L.oracle_pow5):
            f.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_sincosf.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_trace_closest.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_trace_shadow.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_camera_ray.argtypes = [C.c_void_p, C.c_ulonglong, C.c_int, C.c_int, C.c_void_p]
        L.oracle_bsdf_sample.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_ulonglong, C.c_uint, C.c_void_p]
        L.oracle_bsdf_eval.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build_bvh(points, mesh, max_leaf_size):
    """buildBVH (main.cu:20-233) on raw arrays: points float4[n] bytes, mesh Triangle(80 B)[n] bytes.
    Returns (nodes uint8[48*n_nodes], indices int32[n_tris], stats dict)."""
    pts = np.ascontiguousarray(points).view(np.uint8).reshape(-1)
    m = np.ascontiguousarray(mesh).view(np.uint8).reshape(-1)
    n = m.size // 80
    nodes = np.zeros(max(2 * n - 1, 1) * 48, np.uint8)
    idx = np.zeros(n, np.int32)
    st = np.zeros(4, np.int32)
    k = lib().oracle_build_bvh(_p(pts), pts.size // 16, _p(m), n, int(max_leaf_size), _p(nodes), _p(idx), _p(st))
    return nodes[:k * 48].copy(), idx, dict(n_nodes=int(st[0]), largest_leaf=int(st[1]), backups=int(st[2]), depth=int(st[3]))


class OracleScene:
    """The scene exactly as the reference's initRender assembles it (main.cu:235-557)."""

    def __init__(self, config_path=None, base_dir=None, render_number=0, arrays=None):
        if arrays is not None:
            # arrays: dict of uint8 buffers in the reference's layouts (points, normals, uvs, mesh, lights, bvh, indices, materials)
            a = {k: np.ascontiguousarray(v).view(np.uint8) for k, v in arrays.items()}
            self._keep = a
            self.h = lib().oracle_scene_from_arrays(_p(a["points"]), a["points"].size // 16, _p(a["normals"]), a["normals"].size // 16,
                                                    _p(a["uvs"]), a["uvs"].size // 8, _p(a["mesh"]), a["mesh"].size // 80,
                                                    _p(a["lights"]), a["lights"].size // 80, _p(a["bvh"]), a["bvh"].size // 48,
                                                    _p(a["indices"]), _p(a["materials"]), a["materials"].size // 176,
                                                    _p(a["textures"]) if "textures" in a and a["textures"].size else None,
                                                    a["textures"].size // 16 if "textures" in a else 0)
        else:
            base = base_dir if base_dir is not None else os.path.dirname(os.path.abspath(config_path))
            self.h = lib().oracle_scene_load(config_path.encode(), base.encode(), render_number)
        if not self.h:
            raise RuntimeError("oracle: could not load " + str(config_path))
        info = np.zeros(16, np.int32)
        lib().oracle_scene_info(self.h, _p(info))
        self.info = dict(zip(INFO_KEYS, (int(v) for v in info)))

    def close(self):
        if self.h:
            lib().oracle_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def array(self, what):
        if what == "textures":
            buf = np.zeros(lib().oracle_scene_texels(self.h) * 16, np.uint8)
            lib().oracle_scene_get(self.h, 9, _p(buf))
            return buf
        spec = {"points": (0, "n_points", 16), "normals": (1, "n_normals", 16), "uvs": (2, "n_uvs", 8),
                "mesh": (3, "n_tris", 80), "lights": (4, "n_lights", 80), "bvh": (5, "n_nodes", 48),
                "indices": (6, "n_tris", 4), "materials": (7, "n_mats", 176)}[what]
        buf = np.zeros(self.info[spec[1]] * spec[2], np.uint8)
        lib().oracle_scene_get(self.h, spec[0], _p(buf))
        return buf

    def camera(self):
        buf = np.zeros(112, np.uint8)
        lib().oracle_scene_get(self.h, 8, _p(buf))
        return buf

    def render(self, camera=None, width=None, height=None, spp=None, max_depth=None, integrator=None, use_mis=True,
               seed=103033, rect=None, counters=False, threads=1, colors=None):
        """launch_[naive_]unidirectional: returns (sum-of-samples float32 [h,w,4], counters or None, seconds)."""
        i = self.info
        w = width or i["width"]; h = height or i["height"]
        spp = i["spp"] if spp is None else spp
        md = i["max_depth"] if max_depth is None else max_depth
        integ = i["integrator"] if integrator is None else integrator
        cam = self.camera() if camera is None else camera
        x0, y0, x1, y1 = rect or (0, 0, w, h)
        col = np.zeros((h, w, 4), np.float32) if colors is None else colors
        cnt = np.zeros((h, w, 8), np.uint32) if counters else None
        secs = lib().oracle_render(self.h, _p(cam), w, h, spp, md, integ, int(use_mis), seed, x0, y0, x1, y1,
                                   _p(col), _p(cnt) if counters else None, threads)
        return col, cnt, secs

    def trace_closest(self, rays, brute_force=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = len(rays)
        oi = np.zeros((n, 4), np.int32); of = np.zeros((n, 12), np.float32); cnt = np.zeros(8, np.uint32)
        lib().oracle_trace_closest(self.h, n, _p(rays), int(brute_force), _p(oi), _p(of), _p(cnt))
        return oi, of, dict(zip(COUNTER_KEYS, (int(v) for v in cnt)))

    def trace_shadow(self, rays, max_t):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        max_t = np.ascontiguousarray(max_t, np.float32)
        n = len(rays)
        of = np.zeros((n, 3), np.float32); cnt = np.zeros(8, np.uint32)
        lib().oracle_trace_shadow(self.h, n, _p(rays), _p(max_t), _p(of), _p(cnt))
        return of, dict(zip(COUNTER_KEYS, (int(v) for v in cnt)))

    def bsdf_sample(self, material, wi, backface=False, eta_i=1.0, eta_t=1.0, seed=103033, subseq=0):
        wi = np.ascontiguousarray(wi, np.float32)
        out = np.zeros(8, np.float32)
        lib().oracle_bsdf_sample(self.h, material, _p(wi), int(backface), eta_i, eta_t, seed, subseq, _p(out))
        return out

    def bsdf_eval(self, material, wi, wo, eta_i=1.0, eta_t=1.0):
        wi = np.ascontiguousarray(wi, np.float32); wo = np.ascontiguousarray(wo, np.float32)
        out = np.zeros(4, np.float32)
        lib().oracle_bsdf_eval(self.h, material, _p(wi), _p(wo), eta_i, eta_t, _p(out))
        return out


def make_camera(pinhole, pos, rot, fov, w, h, aperture=0.0, focal_dist=0.0):
    pos = np.ascontiguousarray(pos, np.float32); rot = np.ascontiguousarray(rot, np.float32)
    out = np.zeros(112, np.uint8)
    lib().oracle_make_camera(int(pinhole), _p(pos), _p(rot), fov, aperture, focal_dist, w, h, _p(out))
    return out


def camera_ray(camera, x, y, seed=103033):
    out = np.zeros(6, np.float32)
    lib().oracle_camera_ray(_p(camera), seed, x, y, _p(out))
    return out


def finalise(colors, spp):
    c = np.ascontiguousarray(colors, np.float32).copy()
    lib().oracle_finalise(_p(c), c.size // 4, spp)
    return c


def save_bmp(path, rgba, post_process=True):
    """Image::saveImageBMP (imageUtil.cu:69-100) on a [h,w,4] float32 frame, y = 0 bottom."""
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[:2]
    if lib().oracle_save_bmp(path.encode(), _p(rgba), w, h, int(post_process)) != 0:
        raise RuntimeError("oracle: could not write " + path)


def save_csv_mono(path, rgba, channel=0):
    """Image::saveImageCSV_MONO (imageUtil.cu:123-142)."""
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[:2]
    if lib().oracle_save_csv_mono(path.encode(), _p(rgba), w, h, int(channel)) != 0:
        raise RuntimeError("oracle: could not write " + path)


def tonemap_gamma(rgba):
    """gammaCorrect(toneMap(.)) per pixel (imageUtil.cu:202-222)."""
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.zeros_like(rgba)
    lib().oracle_tonemap_gamma(_p(rgba), rgba.size // 4, _p(out))
    return out


def xorwow_init(seed, subseq):
    st = np.zeros(6, np.uint32)
    lib().oracle_xorwow_init(seed, subseq, _p(st))
    return st


def xorwow_next(state, n):
    out = np.zeros(n, np.uint32)
    lib().oracle_xorwow_next(_p(state), n, _p(out))
    return out


def xorwow_uniform(state, n):
    out = np.zeros(n, np.float32)
    lib().oracle_xorwow_uniform(_p(state), n, _p(out))
    return out


def xorwow_matrix(k):
    out = np.zeros((160, 5), np.uint32)
    lib().oracle_xorwow_matrix(k, _p(out))
    return out


def sincosf(x):
    x = np.ascontiguousarray(x, np.float32); s = np.zeros_like(x); c = np.zeros_like(x)
    lib().oracle_sincosf(x.size, _p(x), _p(s), _p(c))
    return s, c


def _unary(fn, x):
    x = np.ascontiguousarray(x, np.float32); y = np.zeros_like(x)
    fn(x.size, _p(x), _p(y))
    return y


def expf(x):
    return _unary(lib().oracle_expf, x)


def rsqrtf(x):
    return _unary(lib().oracle_rsqrtf, x)


def pow5(x):
    return _unary(lib().oracle_pow5, x)
