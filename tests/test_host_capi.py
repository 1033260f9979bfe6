"""CPU suite for the host side: the library loads and exports every symbol include/rt_capi.h declares, the loader's
choose_scene restatement (schema, defaults, BVH build, errors), the binary scene form, scene validation, camera, BMP I/O.
No compute entry point is called: there is no GPU here and no CPU fallback to call."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import capi, scenes
from conftest import ROOT, SCENES_DET

REF_SCENES = "/root/reference/src/scenes"


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    header = open(os.path.join(ROOT, "include", "rt_capi.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found in rt_capi.h"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in rt_capi.h but not exported"
    assert declared == set(capi.SYMBOLS)
    assert b"gfx950" in lib.rt_version()


def test_no_gpu_means_loud_failure_not_fallback(flat_scenes):
    if rt.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError) as e:
        rt.Scene(flat_scenes("a"), 0)
    assert e.value.code == capi.RT_EHIP and "no CPU fallback" in str(e.value)


def test_product_package_does_not_touch_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for needle in ("liboracle", "import oracle", "restate.h", "oracle_trace", "oracle_render", "oracle_intersect", "oracle_shadow"):
                    assert needle not in text, f"{f} uses the oracle ({needle})"


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="reference scenes only exist in the build container")
@pytest.mark.parametrize("name", list(SCENES_DET) + ["m"])
def test_json_loader_reproduces_the_committed_flat_scenes(name, scene_dir, tmp_path):
    # the golden .rtscene files were produced by this loader from the reference's JSON; loading the JSON again must give the same bytes
    fs = rt.FlatScene.from_json(f"{REF_SCENES}/{name}.json", REF_SCENES + "/")
    out = tmp_path / f"{name}.rtscene"
    fs.save(str(out))
    assert open(out, "rb").read() == open(scene_dir / f"{name}.rtscene", "rb").read()


def test_loader_schema_defaults_ids_and_bvh(tmp_path):
    scenes.write_bmp24(str(tmp_path / "t.bmp"), np.arange(4 * 2 * 3, dtype=np.uint8).reshape(2, 4, 3))
    text = """{"camera": {}, "objects": [
        {"type": "sphere", "radius": 1, "position": [0, 0, -5], "material": {"diffuse": [1, 0, 0]}},
        {"type": "plane", "position": [0, -1, 0], "normal": [0, 2, 0], "material": {}},
        {"type": "mesh", "material": {"reflective": [0.5, 0.5, 0.5]}, "triangles": [[[0,0,-3],[1,0,-3],[0,1,-3]], [[2,0,-3],[3,0,-3],[2,1,-3]]]},
        {"type": "box", "point1": [1, 1, -4], "point2": [0, 0, -6], "material": {}},
        {"type": "texturemesh", "texture": "t.bmp", "material": {}, "triangles": [[[5,0,-3],[6,0,-3],[5,1,-3]]], "uvCoords": [[[0,0],[1,0],[0,1]]]},
        {"type": "unknown-thing"}],
      "lights": [{"type": "ambient", "color": [1,1,1]}, {"type": "directional", "color": [1,1,1], "direction": [0, -2, 0]},
                 {"type": "spot", "color": [1,1,1], "position": [0,5,0], "direction": [0,-1,0], "cutoff": 30}]}"""
    fs = rt.FlatScene.from_json_text(text, str(tmp_path) + "/")
    i = fs.info
    assert fs.fov == 60.0 and np.array_equal(fs.background, [0, 0, 0])             # raytracer.cpp:24-25 defaults
    assert (i.n_planes, i.n_spheres, i.n_tris, i.n_boxes, i.n_lights) == (1, 1, 3, 1, 3)
    assert i.n_objects == 6                                                          # plane id 0, then sphere, 2 tris, box, 1 tri
    assert i.n_leaf_prims == 5 and i.n_textures == 1 and i.n_texel_bytes == 24
    planes = np.ctypeslib.as_array(C.cast(i.planes, C.POINTER(C.c_int32)), (8,))
    assert planes[6] == 0                                                            # the plane's id
    tris = np.ctypeslib.as_array(C.cast(i.tris, C.POINTER(C.c_float)), (i.n_tris, 16))
    assert np.allclose(np.abs(tris[:, 9:12]), [0, 0, 1])                             # unit normals from cross(p1-p0, p2-p1)
    leaf = np.ctypeslib.as_array(C.cast(i.leaf_prims, C.POINTER(C.c_uint32)), (i.n_leaf_prims,))
    assert (leaf[-1] & 0x80000000) != 0 and ((leaf >> 28) & 7).max() <= 3
    # 5 BVH objects: median split until <= 2 per leaf (bvh.cpp:39) -> leaves of 1-2 objects
    nodes = np.ctypeslib.as_array(C.cast(i.bvh_nodes, C.POINTER(C.c_int32)), (i.n_bvh_nodes, 16))
    counts = nodes[:, 14:16][nodes[:, 12:14] < 0]
    assert counts.sum() == 5 and counts.max() <= 2
    # texel layout: row 0 = top row, RGB
    tex = np.ctypeslib.as_array(C.cast(i.texels, C.POINTER(C.c_uint8)), (24,))
    assert np.array_equal(tex, np.arange(24, dtype=np.uint8))


def test_loader_errors(tmp_path):
    with pytest.raises(rt.RtError) as e:
        rt.FlatScene.from_json(str(tmp_path / "missing.json"))
    assert e.value.code == capi.RT_EIO and "Unable to open scene file" in str(e.value)   # raytracer.cpp:128
    with pytest.raises(rt.RtError) as e:
        rt.FlatScene.from_json_text('{"camera": {}, "objects": [ {"type": "sphere", "radius": 1 } ], "lights": []}')
    assert e.value.code == capi.RT_EPARSE                                                 # missing "position": nlohmann would throw
    with pytest.raises(rt.RtError) as e:
        rt.FlatScene.from_json_text('{"camera": {}, "objects": [,], "lights": []}')
    assert e.value.code == capi.RT_EPARSE
    # a missing texture is silent: EasyBMP leaves a 1x1 white image (EasyBMP.cpp:685-695)
    fs = rt.FlatScene.from_json_text('{"objects": [{"type": "bumpsphere", "radius": 1, "position": [0,0,-3], "bumpmap": "nope.bmp", "bumpdepth": 1, "material": {}}], "lights": []}', str(tmp_path))
    assert fs.info.n_textures == 1 and fs.info.n_texel_bytes == 3
    # an all-plane scene crashes the reference (bvh.cpp:164); here it is an empty tree
    fs = rt.FlatScene.from_json_text('{"objects": [{"type": "plane", "position": [0,0,0], "normal": [0,1,0], "material": {}}], "lights": []}')
    assert fs.info.root_child == -1 and fs.info.root_count == 0 and fs.info.n_bvh_nodes == 0


def test_csg_leaf_limit(tmp_path):
    def chain(n):
        node = {"type": "sphere", "position": [0, 0, -5], "radius": 1}
        for k in range(n - 1):
            node = {"operation": "union", "first": {"type": "sphere", "position": [k, 0, -5], "radius": 1}, "second": node}
        return node
    import json
    ok = dict(chain(16), type="csgobject", material={})
    rt.FlatScene.from_json_text(json.dumps({"objects": [ok], "lights": []}))
    bad = dict(chain(17), type="csgobject", material={})
    with pytest.raises(rt.RtError):
        rt.FlatScene.from_json_text(json.dumps({"objects": [bad], "lights": []}))


def test_binary_scene_round_trip_and_corruption(flat_scenes, scene_dir, tmp_path):
    fs = flat_scenes("k")
    p = tmp_path / "k2.rtscene"
    fs.save(str(p))
    assert open(p, "rb").read() == open(scene_dir / "k.rtscene", "rb").read()
    data = bytearray(open(p, "rb").read())
    (tmp_path / "trunc.rtscene").write_bytes(data[: len(data) // 2])
    with pytest.raises(rt.RtError):
        rt.FlatScene.load(str(tmp_path / "trunc.rtscene"))
    (tmp_path / "magic.rtscene").write_bytes(b"NOTASCENE" + data[9:])
    with pytest.raises(rt.RtError):
        rt.FlatScene.load(str(tmp_path / "magic.rtscene"))
    # element counts are not trusted: one that does not fit the rest of the file (here 2^35 planes, 2^31 planes) is a parse error,
    # not a 64 GB allocation or an exception thrown through the C boundary
    first_count = 8 + 8 + 12 + 4 + 24 + 4 + 4
    import struct
    for huge in (1 << 35, 1 << 31, (1 << 64) - 1):
        bad = bytearray(data)
        bad[first_count:first_count + 8] = struct.pack("<Q", huge)
        (tmp_path / "count.rtscene").write_bytes(bad)
        with pytest.raises(rt.RtError) as err:
            rt.FlatScene.load(str(tmp_path / "count.rtscene"))
        assert err.value.code == rt.capi.RT_EPARSE


def test_moved_and_turned_cameras_match_the_reference():
    """rt_camera_setup and the primary targets against the compiled reference's setFacing() / s(x, y) (q1.cpp:30-53) for twenty cameras:
    eye positions, rotations about both axes, angles q1.cpp snaps to zero (tests/golden/ref_cameras.npz, tools/make_camera_golden.py)."""
    import os
    import oracle
    from conftest import GOLD
    g = np.load(os.path.join(GOLD, "ref_cameras.npz"))
    for k, c in enumerate(g["cameras"]):
        for W, H in ((16, 9), (7, 13)):
            cam = rt.Camera(float(g["fov"]), W, H, eye=tuple(c[:3]), rot_x=float(c[3]), rot_y=float(c[4]))
            for field in ("eye", "facing", "right", "up"):
                assert np.array_equal(np.array(getattr(cam.c, field)[:], np.float32).view(np.uint32), g[f"{field}_{k}_{W}x{H}"].view(np.uint32)), (k, field)
            assert np.array_equal(oracle.primary_targets(cam, W, H).view(np.uint32), g[f"s_{k}_{W}x{H}"].view(np.uint32)), k


def test_camera_matches_reference_dump(golden):
    g = golden("i")          # fov 75
    cam = rt.Camera(float(g["fov"]), int(g["W"]), int(g["H"]))
    for field in ("facing", "right", "up"):
        assert np.array_equal(np.array(getattr(cam.c, field)[:], np.float32), g[field]), field
    g = golden("l")          # fov 17
    cam = rt.Camera(float(g["fov"]), int(g["W"]), int(g["H"]))
    assert np.array_equal(np.array(cam.c.right[:], np.float32), g["right"])


def test_bmp_write_read_and_quantiser(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.random((5, 7, 3), dtype=np.float32) * 1.4 - 0.2
    img[0, 0] = np.nan
    q = rt.quantize_rgb8(img)
    exp = (np.clip(np.nan_to_num(img, nan=0.0), 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8)[::-1]
    assert np.array_equal(q, exp)                       # clamp, *255+0.5, NaN -> 0, rows flipped (row 0 of the frame is the bottom)
    path = tmp_path / "o.bmp"
    rt.write_bmp(str(path), img)
    raw = open(path, "rb").read()
    assert raw[:2] == b"BM" and len(raw) == 54 + 5 * 24  # 7*3 = 21 bytes per row padded to 24
    # read it back through the loader's BMP reader: use it as a texture
    fs = rt.FlatScene.from_json_text('{"objects": [{"type": "bumpsphere", "radius": 1, "position": [0,0,-3], "bumpmap": "o.bmp", "bumpdepth": 1, "material": {}}], "lights": []}', str(tmp_path))
    tex = np.ctypeslib.as_array(C.cast(fs.info.texels, C.POINTER(C.c_uint8)), (5 * 7 * 3,)).reshape(5, 7, 3)
    assert np.array_equal(tex, q)


def test_synthetic_config_scenes_load_and_match_oracle_counts(tmp_path):
    import oracle
    for fn, kw in ((scenes.c2_scene, dict(grid=24)), (scenes.c3_scene, {}), (scenes.c4_scene, dict(samples=4)), (scenes.c5_scene, dict(grid=16))):
        fs = rt.FlatScene.from_json(fn(str(tmp_path), **kw))
        cam = rt.Camera(fs.fov, 48, 27)
        img, nn, ns = oracle.render(fs, cam, 48, 27, max_depth=5, seed=3)
        assert nn >= 48 * 27 and np.isfinite(img).all() and img.max() > 0.05


def test_binary_mesh_file_loads_to_the_same_scene_as_json(tmp_path):
    """"file": mesh.bin is an additive form of "triangles"/"uvCoords"; both give bit-identical flat scenes."""
    from raytracer_amd import scenes
    a = rt.FlatScene.from_json(scenes.c5_scene(str(tmp_path), grid=40, binary=False))
    b = rt.FlatScene.from_json(scenes.c5_scene(str(tmp_path), grid=40, binary=True))
    a.save(str(tmp_path / "a.rtscene")); b.save(str(tmp_path / "b.rtscene"))
    assert (tmp_path / "a.rtscene").read_bytes() == (tmp_path / "b.rtscene").read_bytes()
    # textured meshes carry their uv block
    import json, numpy as np
    tris, uvs = scenes.height_field(6, seed=3)
    scenes.write_bmp24(str(tmp_path / "t.bmp"), scenes.tiles_texture(64, 4))
    base = {"camera": {"field": 60, "background": [0, 0, 0]}, "lights": [{"type": "ambient", "color": [1, 1, 1]}]}
    mj = {"type": "texturemesh", "texture": "t.bmp", "material": {}}
    mj.update(scenes._mesh_json(tris, uvs))
    mf = {"type": "texturemesh", "texture": "t.bmp", "material": {}}
    mf.update(scenes._mesh_file(str(tmp_path), "m.bin", tris, uvs))
    for name, mesh in (("j", mj), ("f", mf)):
        (tmp_path / f"{name}.json").write_text(json.dumps(dict(base, objects=[mesh])))
        rt.FlatScene.from_json(str(tmp_path / f"{name}.json")).save(str(tmp_path / f"{name}.rtscene"))
    assert (tmp_path / "j.rtscene").read_bytes() == (tmp_path / "f.rtscene").read_bytes()
    # a missing or truncated file is an error, not a crash
    (tmp_path / "bad.json").write_text(json.dumps(dict(base, objects=[{"type": "mesh", "material": {}, "file": "nope.bin"}])))
    with pytest.raises(rt.capi.RtError):
        rt.FlatScene.from_json(str(tmp_path / "bad.json"))
    (tmp_path / "short.bin").write_bytes((tmp_path / "m.bin").read_bytes()[:100])
    (tmp_path / "short.json").write_text(json.dumps(dict(base, objects=[{"type": "mesh", "material": {}, "file": "short.bin"}])))
    with pytest.raises(rt.capi.RtError):
        rt.FlatScene.from_json(str(tmp_path / "short.json"))


def build_shim_client(tmp_path, extra=()):
    """tests/shim_client.cpp (a q1.cpp-style caller of the reference API) against include/raytracer.h + libraytracer_amd.so."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "shim_client")
    lib_dir = os.path.join(ROOT, "raytracer_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", *extra, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "shim_client.cpp"), "-o", exe,
                    "-L" + lib_dir, "-lraytracer_amd", "-Wl,-rpath," + lib_dir], check=True, capture_output=True)
    return exe


def test_reference_api_header_with_the_reference_glm_types(tmp_path):
    """With RT_USE_GLM, point3 / colour3 are glm::vec3 as in the reference (raytracer.h:9-10): the same caller builds against
    the reference's vendored GLM (only where the reference tree is present: the build container)."""
    if not os.path.isdir("/root/reference/glm"):
        pytest.skip("the reference tree is not here")
    build_shim_client(tmp_path, extra=("-DRT_USE_GLM", "-I/root/reference/glm", "-w"))


def test_reference_api_header_compiles_and_links(tmp_path):
    """The drop-in boundary: a caller of choose_scene / trace / shadowRay / fov / background_colour builds against this
    repository's raytracer.h and resolves every rtapi_* symbol in the library (it is not run here: no GPU)."""
    build_shim_client(tmp_path)
    lib = capi.load()
    header = open(os.path.join(ROOT, "include", "raytracer.h")).read()
    for name in set(re.findall(r"\b(rtapi_[a-z0-9_]+)\s*\(", header)):
        assert hasattr(lib, name), f"{name} is declared in raytracer.h but not exported"


def test_reference_api_missing_scene_exits_like_the_reference(tmp_path):
    """choose_scene on a scene file that does not exist: the reference prints "Unable to open scene file ..." and calls
    exit(EXIT_FAILURE) (raytracer.cpp:127-130); so does the stand-in, before anything touches a GPU."""
    import subprocess
    exe = build_shim_client(tmp_path)
    r = subprocess.run([exe, "no_such_scene", "16", "16"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 1
    assert "Loading scene no_such_scene" in r.stdout and "Unable to open scene file scenes/no_such_scene.json" in r.stdout
