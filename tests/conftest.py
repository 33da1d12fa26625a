import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hanamaru-renderer_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the g++-only parts (host layer, oracle, emulation) are built on demand; the HIP library is built by
    # __graft_entry__.build() and travels to the GPU box prebuilt.
    import __graft_entry__ as ge
    ge.build_cpu_parts()   # `make`: a no-op when the libraries are up to date, a rebuild when a source is newer


@pytest.fixture(scope="session")
def ha():
    import hanamaru_amd
    return hanamaru_amd


@pytest.fixture(scope="session")
def orc():
    import oracle_py
    return oracle_py


@pytest.fixture(scope="session")
def emu():
    import emu_py
    return emu_py


_scene_cache = {}


@pytest.fixture(scope="session")
def scenes(ha, orc):
    """name -> (host scene, oracle scene), built lazily and cached for the session."""
    def get(name):
        if name not in _scene_cache:
            sc = ha.Scene(name)
            _scene_cache[name] = (sc, orc.OracleScene(sc.desc_ptr))
        return _scene_cache[name]
    return get


@pytest.fixture(scope="session")
def gpu(ha):
    """A Renderer on cuda:0.  Fails loudly when the HIP library or the device is missing."""
    r = ha.Renderer(0)
    yield r
    r.close()
