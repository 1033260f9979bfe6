"""GPU suite, part 5: degenerate scenes and empty inputs against the oracle restatement -- no objects at all, planes only (the
reference itself crashes on that one, bvh.cpp:164: here it is an empty tree), a tree that is a single leaf of one or two
primitives, no lights, two mirrors facing each other (every ray runs into the depth guard of raytracer.cpp:308-313)."""
import json

import numpy as np
import pytest

import oracle
import raytracer_amd as rt

pytestmark = pytest.mark.gpu

AMBIENT = {"type": "ambient", "color": [0.2, 0.2, 0.2]}
POINT = {"type": "point", "color": [0.9, 0.8, 0.7], "position": [1, 4, 1]}
DIRECTIONAL = {"type": "directional", "color": [0.5, 0.5, 0.6], "direction": [0.3, -1, -0.4]}
MATT = {"ambient": [0.3, 0.2, 0.1], "diffuse": [0.6, 0.5, 0.4], "specular": [0.4, 0.4, 0.4], "shininess": 12}
MIRROR = dict(MATT, reflective=[0.9, 0.9, 0.9])
FLOOR = {"type": "plane", "position": [0, -1, 0], "normal": [0, 1, 0], "material": MATT}

SCENES = {
    "nothing": {"objects": [], "lights": []},
    "nothing_but_lights": {"objects": [], "lights": [AMBIENT, POINT, DIRECTIONAL]},
    "planes_only": {"objects": [FLOOR, {"type": "plane", "position": [0, 0, -9], "normal": [0, 0, 1], "material": MIRROR}], "lights": [AMBIENT, POINT, DIRECTIONAL]},
    "one_sphere_no_lights": {"objects": [{"type": "sphere", "position": [0, 0, -4], "radius": 1, "material": MATT}], "lights": []},
    "one_triangle": {"objects": [{"type": "mesh", "material": MATT, "triangles": [[[-1, -1, -4], [1, -1, -4], [0, 1, -4.5]]]}, FLOOR], "lights": [AMBIENT, DIRECTIONAL]},
    "one_leaf_of_two": {"objects": [{"type": "sphere", "position": [-0.7, 0, -4], "radius": 0.6, "material": MIRROR},
                                    {"type": "box", "point1": [0.2, -0.5, -4.5], "point2": [1.2, 0.5, -3.5], "material": MATT}, FLOOR],
                        "lights": [AMBIENT, POINT, DIRECTIONAL, {"type": "spot", "color": [1, 1, 1], "position": [0, 5, -4], "direction": [0, -1, 0], "cutoff": 20}]},
    "facing_mirrors": {"objects": [{"type": "plane", "position": [0, -1, 0], "normal": [0, 1, 0], "material": MIRROR},
                                   {"type": "box", "point1": [-50, 1, -50], "point2": [50, 2, 50], "material": MIRROR}],
                       "lights": [AMBIENT, POINT]},
}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rgb8(rgb):
    q = np.clip(np.nan_to_num(np.asarray(rgb, np.float32), nan=0.0), 0, 1)
    return (q * np.float32(255) + np.float32(0.5)).astype(np.uint8)


@pytest.mark.parametrize("name", list(SCENES))
def test_degenerate_scene_vs_oracle(name):
    desc = dict(SCENES[name], camera={"field": 60, "background": [0.1, 0.2, 0.3]})
    fs = rt.FlatScene.from_json_text(json.dumps(desc))
    sc = rt.Scene(fs, 0)
    try:
        W, H = 96, 64
        cam = rt.Camera(fs.fov, W, H)
        s = oracle.primary_targets(cam, W, H)
        e = np.zeros_like(s)
        r = sc.trace_batch(e, s)
        st = dict(sc.last_stats)
        o = oracle.trace_batch(fs, e, s, seed=1, threads=4)
        assert np.array_equal(r["hit"], o["hit"]) and np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
        assert st["rays_nearest"] == o["rays_nearest"] and st["rays_shadow"] == o["rays_shadow"]
        assert np.abs(rgb8(r["rgb"]).astype(int) - rgb8(o["rgb"]).astype(int)).max() <= 1
        img = sc.render(W, H, camera=cam)
        ref, nn, ns = oracle.render(fs, cam, W, H, seed=1, threads=4)
        assert np.abs(rgb8(img).astype(int) - rgb8(ref).astype(int)).max() <= 1
        assert sc.last_stats["rays_nearest"] == nn and sc.last_stats["rays_shadow"] == ns
        if name.startswith("nothing"):
            assert not r["hit"].any() and np.array_equal(img, np.broadcast_to(np.float32([0.1, 0.2, 0.3]), img.shape))
        if name == "facing_mirrors":
            assert st["rays_nearest"] > 10 * W * H * 0.4          # the rays between the mirrors run to the depth limit
        # shadow queries and nearest-hit queries of arbitrary rays
        rng = np.random.default_rng(5)
        p = rng.uniform(-3, 3, (4096, 3)).astype(np.float32)
        q = rng.uniform(-6, 6, (4096, 3)).astype(np.float32)
        a, b = sc.shadow_batch(p, q), oracle.shadow_batch(fs, p, q)
        assert np.array_equal(a["lit"], b["lit"])
        lit = a["lit"] != 0
        assert np.allclose(a["shadow"][lit], b["shadow"][lit], rtol=1e-6, atol=0)
        a, b = sc.intersect_batch(p, q - p), oracle.intersect_batch(fs, p, q - p)
        assert np.array_equal(a["hit"], b["hit"])
        hit = a["hit"] != 0
        assert np.array_equal(a["id"][hit], b["id"][hit]) and np.array_equal(bits(a["t"][hit]), bits(b["t"][hit]))
    finally:
        sc.close()


def test_empty_batches(gpu_scenes):
    sc = gpu_scenes("c")
    none = np.zeros((0, 3), np.float32)
    r = sc.trace_batch(none, none)
    assert r["hit"].shape == (0,) and r["rgb"].shape == (0, 3) and sc.last_stats["rays_nearest"] == 0
    assert sc.intersect_batch(none, none)["id"].shape == (0,)
    assert sc.shadow_batch(none, none)["lit"].shape == (0,)
    # NULL arrays are fine for n = 0 at the C-ABI, not for n > 0
    import ctypes as C
    from raytracer_amd import capi
    lib = capi.load()
    assert lib.rt_trace_batch(sc._h, None, None, 0, 16, 1, 0, None, None, None, None, None) == capi.RT_OK
    assert lib.rt_intersect_batch(sc._h, None, None, 0, 0, None, None, None, None) == capi.RT_OK
    assert lib.rt_shadow_batch(sc._h, None, None, 0, 0, None, None, None) == capi.RT_OK
    assert lib.rt_trace_batch(sc._h, None, None, 5, 16, 1, 0, None, None, None, None, None) == capi.RT_EINVAL
