import gzip
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")
SCENES_DET = "abcdefghijkl"     # deterministic bundled scenes; m has rand()-driven area lights
# the compiled reference on scenes of raytracer_amd.scenes.random_scene (tools/make_random_golden.py): everything at once, 96x64
SCENES_RANDOM = ("random_5", "random_8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scene_dir(tmp_path_factory):
    """The golden scenes (flat binary form, see tools/make_golden.py) unpacked into a scratch directory."""
    d = tmp_path_factory.mktemp("scenes")
    for f in os.listdir(os.path.join(GOLD, "scenes")):
        if f.endswith(".gz"):
            with gzip.open(os.path.join(GOLD, "scenes", f), "rb") as src, open(d / f[:-3], "wb") as dst:
                shutil.copyfileobj(src, dst)
    return d


@pytest.fixture(scope="session")
def flat_scenes(scene_dir):
    import raytracer_amd as rt
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = rt.FlatScene.load(str(scene_dir / f"{name}.rtscene"))
        return cache[name]
    return get


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLD, f"ref_{name}.npz")))
        return cache[name]
    return get


@pytest.fixture(scope="session")
def gpu_scenes(flat_scenes):
    """Scenes uploaded to cuda:0, cached for the session (one rt_scene per scene)."""
    import raytracer_amd as rt
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = rt.Scene(flat_scenes(name), 0)
        return cache[name]
    yield get
    for s in cache.values():
        s.close()
