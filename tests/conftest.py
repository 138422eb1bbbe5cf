"""Shared fixtures. GPU tests are marked `gpu`; everything else runs on CPU.

Only tests (and bench.py's cpu_baseline leg / smoke()) touch oracle/ — see oracle/README.md.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def api():
    from cudapathtracer_amd import api as A
    A.lib()
    return A


@pytest.fixture(scope="session")
def scene_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("scenes"))


def golden_cases():
    """Names of the small full-frame fixtures (tests/golden/*.npz without the window_* files)."""
    import glob
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith("window_"))


def window_cases():
    """Names of the window fixtures: 64x64 windows of the real BASELINE scenes at the 1080p camera."""
    import glob
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "window_*.npz")))


def window_scene(g, outdir):
    """Regenerate the scene of a window fixture from scenes.py and check it is the scene the fixture was made from."""
    from cudapathtracer_amd import scenes
    s = getattr(scenes, str(g["generator"]))(outdir, width=int(g["w"]), height=int(g["h"]), spp=int(g["spp"]), max_depth=int(g["max_depth"]),
                                              name=os.path.basename(outdir))
    return s


def golden_scene(name, sub="scenes"):
    return os.path.join(GOLDEN, sub, name + ".rendertron")


def golden_case_scene(g):
    """Config path of a golden .npz (older fixtures have no scene_dir field: they live in scenes/)."""
    return golden_scene(str(g["scene"]), str(g["scene_dir"]) if "scene_dir" in g else "scenes")


@pytest.fixture(scope="session")
def gpu_ready(api):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no HIP device is visible")
    torch.cuda.set_device(0)
    return torch
