import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "tinyraytracing_amd", "lib", "libtrt_host.so"),
            os.path.join(ROOT, "tinyraytracing_amd", "lib", "libtrt_hip.so"),
            os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "oracle", "liboracle_exp.so"),
            os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")]
    if all(os.path.exists(p) for p in need):
        return
    import __graft_entry__ as g
    g.build()


_ensure_built()

import tinyraytracing_amd as T  # noqa: E402

_scene_cache = {}


def get_scene(name, width, height, **kw):
    key = (name, width, height, tuple(sorted(kw.items())))
    if key not in _scene_cache:
        _scene_cache[key] = T.Scene.named(name, width, height, **kw)
    return _scene_cache[key]


@pytest.fixture(scope="session")
def scene_factory():
    return get_scene


_renderers = {}


@pytest.fixture(scope="session")
def renderer_factory():
    """GPU tests only: one Renderer (scene resident in HBM) per scene."""
    def make(scene):
        k = id(scene)
        if k not in _renderers:
            _renderers[k] = T.Renderer(scene, 0)
        return _renderers[k]
    return make
