import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a MI355X (run with -m gpu)")


def pkg():
    return importlib.import_module("raytracing-project_amd")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def gpu_scene_from_oracle(s, device=0):
    """RtScene built from the oracle's own scene arrays, so that a parity test
    compares the two tracers on identical inputs."""
    capi = pkg().capi
    vp, vn = s.vertices()
    tv, tt, tm = s.triangles()
    nd = s.nodes()
    return capi.Scene(vp, vn, tv, tt, tm, s.texcoords(), s.materials(), s.textures(),
                      nd["bounds"], nd["left"], nd["right"], nd["tri_off"], nd["tri_cnt"],
                      nd["tris"], device=device)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
