"""The drop-in boundary: libptamd.so loads on a machine without a GPU and exports every symbol
include/pt_api.h declares; boundary structs keep the reference's CUDA layouts."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b((?:pt|novum)_[a-z_0-9]+)\s*\(", txt)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(api):
    names = _declared_functions()
    assert len(names) >= 30
    L = api.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_version_and_layouts(api):
    L = api.lib()
    assert L.pt_api_version() == 1
    assert ctypes.sizeof(api.Camera) == 112 and api.Camera.forward.offset == 64 and api.Camera.fovScale.offset == 44
    assert ctypes.sizeof(api.TileRange) == 12


def test_errors_do_not_throw_across_the_abi(api):
    L = api.lib()
    assert L.pt_scene_create(None) is None
    assert b"null desc" in L.pt_last_error()
    assert L.pt_get_counters(None, None) != 0


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "cudapathtracer_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                assert "oracle" not in open(os.path.join(d, f)).read().lower().replace("oracle/readme", ""), os.path.join(d, f)


def test_tile_ranges(api):
    assert api.n_tiles(1920, 1080) == 240 * 135 and api.n_tiles(33, 9) == 5 * 2
    for world in (1, 2, 3, 8):
        covered = []
        for r in range(world):
            tr = api.rank_tiles(70, 41, r, world)
            covered += [tr.first + k * tr.stride for k in range(tr.count)]
        assert sorted(covered) == list(range(api.n_tiles(70, 41)))
