"""GPU suite, part 2: the BASELINE.json configurations at their real sizes.  The full-size C2 scene has a dump of the compiled
reference at 256x144 (tests/golden/ref_c2_224.npz); whole 1080p frames are beyond the reference and the oracle, so they are pinned by
  * the oracle on sampled rays of the full-size frame (ids exact, t bit-exact, 8-bit RGB),
  * size-independent properties: the frame is bit-identical run to run, any tile sharding composes to the same frame, and
    the two traversal forms (packets for camera rays, per-lane walks for caller-supplied rays) give bit-identical images,
  * ray counts equal to the oracle's BVH::findNearest / BVH::calcShadow counts on a reduced frame."""
import json
import os

import numpy as np
import pytest

import oracle
import raytracer_amd as rt
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rgb8(rgb):
    q = np.clip(np.nan_to_num(np.asarray(rgb, np.float32), nan=0.0), 0, 1)
    return (q * np.float32(255) + np.float32(0.5)).astype(np.uint8)


def primary_targets_np(cam, W, H):
    """q1.cpp:51-53 in float32 numpy, same operation order as the kernels and the oracle (checked below)."""
    f = np.float32
    eye, facing, right, up = (np.array(v[:], f) for v in (cam.c.eye, cam.c.facing, cam.c.right, cam.c.up))
    x = np.arange(W, dtype=f)[None, :, None]
    y = np.arange(H, dtype=f)[:, None, None]
    a = f(2) * ((x + f(0.5)) / f(W) - f(0.5))
    b = f(2) * ((y + f(0.5)) / f(H) - f(0.5))
    s = ((eye + facing)[None, None, :] + right[None, None, :] * a) + up[None, None, :] * b
    return s.reshape(-1, 3).astype(f)


@pytest.fixture(scope="module")
def c2(tmp_path_factory):
    flat = rt.FlatScene.from_json(scenes.c2_scene(str(tmp_path_factory.mktemp("c2"))))
    sc = rt.Scene(flat, 0)
    yield flat, sc
    sc.close()


def test_numpy_primary_targets_equal_oracle(c2):
    flat, _ = c2
    cam = rt.Camera(flat.fov, 96, 54, eye=(0, 1.5, 1.0), rot_x=-0.3)
    assert np.array_equal(bits(primary_targets_np(cam, 96, 54)), bits(oracle.primary_targets(cam, 96, 54)))


def test_c2_full_size_vs_compiled_reference(c2):
    """The benchmark's own scene -- all 100 352 triangles, generated again here by the same deterministic generator -- against what
    the COMPILED REFERENCE computed for it at 256x144 from bench.py's camera (tools/make_random_golden.py c2_224): ids, t and ray
    counts exact, RGB within 1/255 (texel-boundary pixels on the bump-mapped spheres counted, as below)."""
    import os
    from conftest import GOLD
    flat, sc = c2
    g = dict(np.load(os.path.join(GOLD, "ref_c2_224.npz")))
    n = int(g["W"]) * int(g["H"])
    e = np.tile(g["eye"].astype(np.float32), (n, 1))
    r = sc.trace_batch(e, g["s"])
    assert np.array_equal(r["hit"], g["hit"]) and np.array_equal(r["id"], g["id"]) and np.array_equal(bits(r["t"]), bits(g["t"]))
    assert sc.last_stats["rays_nearest"] == int(g["rays_nearest"]) and sc.last_stats["rays_shadow"] == int(g["rays_shadow"])
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(g["rgb"]).astype(int)).max(axis=1)
    # the bump-mapped spheres pick their texels with glibc's atan2f / asinf restated on the device (glibc_math.h): no pixel is off
    assert int((d8 > 1).sum()) == 0, f"{int((d8 > 1).sum())} of {n} pixels beyond 1/255"
    a = sc.shadow_batch(g["shd_p"], g["shd_l"])
    assert np.array_equal(a["lit"], g["shd_lit"].astype(np.uint8))
    # the frame path (packets, lanes) on the same camera
    cam = rt.Camera(flat.fov, int(g["W"]), int(g["H"]), eye=tuple(float(v) for v in g["eye"]), rot_x=-0.3)
    assert np.array_equal(bits(primary_targets_np(cam, int(g["W"]), int(g["H"]))), bits(g["s"]))
    img = sc.render(int(g["W"]), int(g["H"]), camera=cam)
    assert np.array_equal(bits(img.reshape(-1, 3)), bits(r["rgb"]))


def test_c2_1080p_sampled_rays_vs_oracle(c2):
    flat, sc = c2
    W, H = 1920, 1080
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    s = primary_targets_np(cam, W, H)
    pick = np.random.default_rng(5).choice(W * H, 30000, replace=False)
    e = np.tile(np.array(cam.c.eye[:], np.float32), (len(pick), 1))
    r = sc.trace_batch(e, s[pick])
    o = oracle.trace_batch(flat, e, s[pick], seed=1, threads=8)
    assert np.array_equal(r["hit"], o["hit"]) and np.array_equal(r["id"], o["id"])
    assert np.array_equal(bits(r["t"]), bits(o["t"]))
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(o["rgb"]).astype(int)).max(axis=1)
    assert int((d8 > 1).sum()) == 0, f"{(d8 > 1).sum()} of {len(pick)} sampled pixels differ by more than 1/255"


def test_c2_1080p_frame_properties(c2):
    import torch
    flat, sc = c2
    W, H = 1920, 1080
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    a = sc.render(W, H, camera=cam)
    st = dict(sc.last_stats)
    b = sc.render(W, H, camera=cam)
    assert np.array_equal(bits(a), bits(b))                                   # deterministic
    full = torch.zeros(H, W, 3, device="cuda")
    for rank in range(4):
        sc.render_into(full.data_ptr(), W, H, camera=cam, tile_rank=rank, tile_ranks=4)
    assert np.array_equal(bits(full.cpu().numpy()), bits(a))                  # tile shards compose
    # packets (camera rays) and per-lane walks (caller-supplied rays) agree bit for bit over the whole frame
    s = primary_targets_np(cam, W, H)
    e = np.tile(np.array(cam.c.eye[:], np.float32), (W * H, 1))
    r = sc.trace_batch(e, s)
    assert np.array_equal(bits(r["rgb"].reshape(H, W, 3)), bits(a))
    assert sc.last_stats["rays_nearest"] == st["rays_nearest"] == W * H and sc.last_stats["rays_shadow"] == st["rays_shadow"]
    assert st["rays_shadow"] == 2 * int(r["hit"].sum())                       # one shadow ray per hit and non-ambient light


def test_c2_ray_counts_vs_oracle_reduced_frame(c2):
    flat, sc = c2
    W, H = 480, 270
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    img = sc.render(W, H, camera=cam)
    ref, nn, ns = oracle.render(flat, cam, W, H, seed=1, threads=8)
    assert sc.last_stats["rays_nearest"] == nn and sc.last_stats["rays_shadow"] == ns
    d8 = np.abs(rgb8(img).astype(int) - rgb8(ref).astype(int)).max(axis=2)
    assert int((d8 > 1).sum()) == 0


def test_c2_bump_spheres_every_pixel_of_the_1080p_frame_vs_oracle(c2):
    """Every 1080p pixel whose primary ray hits one of the two bump-mapped spheres (the only surfaces whose shading goes through a
    truncated transcendental, bump.cpp:14-23), device against the oracle: the measured number of pixels beyond 1/255 is zero."""
    flat, sc = c2
    W, H = 1920, 1080
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    s = primary_targets_np(cam, W, H)
    e = np.tile(np.array(cam.c.eye[:], np.float32), (W * H, 1))
    ids = sc.intersect_batch(e, s - e)["id"]
    on_sphere = np.flatnonzero(ids >= flat.info.n_objects - 2)          # the two spheres were loaded last
    assert len(on_sphere) > 20000
    r = sc.trace_batch(e[on_sphere], s[on_sphere])
    o = oracle.trace_batch(flat, e[on_sphere], s[on_sphere], seed=1, threads=8)
    assert np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(o["rgb"]).astype(int)).max(axis=1)
    print(f"{len(on_sphere)} bump-sphere pixels at 1080p, {int((d8 > 1).sum())} beyond 1/255, max float difference {np.abs(r['rgb'] - o['rgb']).max():.2e}")
    assert int((d8 > 1).sum()) == 0


def _against_dump(sc, g, rgb_exact=True):
    n = int(g["W"]) * int(g["H"])
    e = np.tile(g["eye"].astype(np.float32), (n, 1))
    r = sc.trace_batch(e, g["s"])
    assert np.array_equal(r["hit"], g["hit"]) and np.array_equal(r["id"], g["id"]) and np.array_equal(bits(r["t"]), bits(g["t"]))
    assert sc.last_stats["rays_nearest"] == int(g["rays_nearest"]) and sc.last_stats["rays_shadow"] == int(g["rays_shadow"])
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(g["rgb"]).astype(int)).max(axis=1)
    assert d8.max() <= 1 and np.nanmax(np.abs(r["rgb"] - g["rgb"])) < 2e-3
    if len(g["sec_id"]):
        a = sc.intersect_batch(g["sec_e"], g["sec_d"])
        assert np.array_equal(a["id"], g["sec_id"]) and np.array_equal(bits(a["t"]), bits(g["sec_t"]))
    a = sc.shadow_batch(g["shd_p"], g["shd_l"])
    assert np.array_equal(a["lit"], g["shd_lit"].astype(np.uint8))
    lit = a["lit"] != 0
    assert np.allclose(a["shadow"][lit], g["shd_shadow"][lit], rtol=1e-6, atol=0)
    return r


def test_fat_leaves_and_lone_meshes_vs_compiled_reference(tmp_path):
    """Two more dumps of the COMPILED REFERENCE (tools/make_random_golden.py c5_300 mesh_3, scenes generated again here):
    a C5-style mesh of 180 000 triangles, which the reference's depth cap of 16 leaves in 65 536 leaves of up to 3 triangles and the
    device walks through conservative sub-trees, and the scene of meshes asked for hits on their own (Mesh::rayhit: a refractive
    mesh, a mesh as CSG operand; per-mesh trees on the device).  192x108 each: ids, t, ray counts exact, RGB within 1/255."""
    import os
    from conftest import GOLD
    for name, make in (("c5_300", lambda d: scenes.c5_scene(d, grid=300, binary=False)), ("mesh_3", lambda d: scenes.mesh_scene(d, subdiv=3))):
        g = dict(np.load(os.path.join(GOLD, f"ref_{name}.npz")))
        flat = rt.FlatScene.from_json(make(str(tmp_path / name)))
        sc = rt.Scene(flat, 0)
        _against_dump(sc, g)
        sc.close()


def test_c3_vs_compiled_reference(tmp_path):
    """BASELINE config 3 against what the COMPILED REFERENCE computed for it at 240x135 (tests/golden/ref_c3.npz, the reference's own
    depth limit of 16): ids, t, ray counts exact, RGB within 1/255; sampled secondary rays and shadow queries."""
    import os
    from conftest import GOLD
    g = dict(np.load(os.path.join(GOLD, "ref_c3.npz")))
    flat = rt.FlatScene.from_json(scenes.c3_scene(str(tmp_path)))
    sc = rt.Scene(flat, 0)
    n = int(g["W"]) * int(g["H"])
    r = sc.trace_batch(np.zeros((n, 3), np.float32), g["s"])
    assert np.array_equal(r["hit"], g["hit"]) and np.array_equal(r["id"], g["id"]) and np.array_equal(bits(r["t"]), bits(g["t"]))
    assert sc.last_stats["rays_nearest"] == int(g["rays_nearest"]) and sc.last_stats["rays_shadow"] == int(g["rays_shadow"])
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(g["rgb"]).astype(int)).max(axis=1)
    assert d8.max() <= 1 and np.nanmax(np.abs(r["rgb"] - g["rgb"])) < 2e-3
    a = sc.intersect_batch(g["sec_e"], g["sec_d"])
    assert np.array_equal(a["id"], g["sec_id"]) and np.array_equal(bits(a["t"]), bits(g["sec_t"]))
    a = sc.shadow_batch(g["shd_p"], g["shd_l"])
    assert np.array_equal(a["lit"], g["shd_lit"].astype(np.uint8))
    lit = a["lit"] != 0
    assert np.allclose(a["shadow"][lit], g["shd_shadow"][lit], rtol=1e-6, atol=0)
    sc.close()


def test_c3_nested_csg_depth5(tmp_path):
    flat = rt.FlatScene.from_json(scenes.c3_scene(str(tmp_path)))
    sc = rt.Scene(flat, 0)
    W, H = 480, 270
    cam = rt.Camera(flat.fov, W, H)
    s = primary_targets_np(cam, W, H)
    e = np.zeros((W * H, 3), np.float32)
    r = sc.trace_batch(e, s, max_depth=5)
    o = oracle.trace_batch(flat, e, s, max_depth=5, seed=1, threads=8)
    assert np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
    assert sc.last_stats["rays_nearest"] == o["rays_nearest"] and sc.last_stats["rays_shadow"] == o["rays_shadow"]
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(o["rgb"]).astype(int)).max(axis=1)
    assert (d8 > 1).mean() < 1e-3, f"{(d8 > 1).sum()} pixels differ by more than 1/255"
    # full size: properties only
    a = sc.render(1920, 1080, max_depth=5)
    b = sc.render(1920, 1080, max_depth=5)
    assert np.array_equal(bits(a), bits(b)) and sc.last_stats["levels"] == 6
    sc.close()


@pytest.mark.parametrize("seed,max_leaves", [(1, 5), (2, 16), (3, 16)])
def test_random_csg_programs(tmp_path, seed, max_leaves):
    """Random operator trees up to the 16-leaf limit (multi-interval lists, empty operands): the interval evaluation in all
    its homes -- packets, per-lane walks, shadow rays, the exit search of transmitted rays -- against the oracle."""
    flat = rt.FlatScene.from_json(scenes.random_csg_scene(str(tmp_path), seed=seed, max_leaves=max_leaves))
    sc = rt.Scene(flat, 0)
    W, H = 320, 180
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.0, 1.5), rot_x=-0.2)
    s = primary_targets_np(cam, W, H)
    e = np.tile(np.array(cam.c.eye[:], np.float32), (W * H, 1))
    r = sc.trace_batch(e, s, max_depth=4)
    o = oracle.trace_batch(flat, e, s, max_depth=4, seed=1, threads=8)
    assert np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
    assert sc.last_stats["rays_nearest"] == o["rays_nearest"] and sc.last_stats["rays_shadow"] == o["rays_shadow"]
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(o["rgb"]).astype(int)).max(axis=1)
    assert (d8 > 1).mean() < 1e-3, f"{(d8 > 1).sum()} pixels differ by more than 1/255"
    img = sc.render(W, H, camera=cam, max_depth=4)          # the frame path (packets) agrees with the batch path
    assert np.array_equal(bits(img.reshape(-1, 3)), bits(r["rgb"]))
    sc.close()


@pytest.mark.parametrize("seed", [0, 12, 19, 38])
def test_random_mixed_scenes(tmp_path, seed):
    """The fuzzer's scenes (tools/fuzz_parity.py ran 160 seeds on the MI355X: all agree): every primitive, material and
    light kind at once, depth 6.  Seeds 12 and 19 hold shadow rays through three transmissive occluders, whose colours
    multiply in visiting order: frame and batch path may differ there in the last bit."""
    flat = rt.FlatScene.from_json(scenes.random_scene(str(tmp_path), seed=seed))
    sc = rt.Scene(flat, 0)
    W, H = 256, 144
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.0, 1.5), rot_x=-0.2)
    s = primary_targets_np(cam, W, H)
    e = np.tile(np.array(cam.c.eye[:], np.float32), (W * H, 1))
    r = sc.trace_batch(e, s, max_depth=6, seed=7)
    # A reflective textured triangle that its own reflected ray's subtree comes back to is lit by the reference with the texel of the
    # LAST hit on it (texturemesh.cpp:49-50 writes the triangle's material, objects.cpp:77-79 reads it after the recursion).  The
    # oracle reproduces that, and so does the device (k_texel_fix); the oracle with that state switched off shows how many pixels
    # of the scene it concerns.
    o = oracle.trace_batch(flat, e, s, max_depth=6, seed=7, threads=8)
    with oracle.own_texels():
        own = oracle.trace_batch(flat, e, s, max_depth=6, seed=7, threads=8)
    state = np.abs(own["rgb"] - o["rgb"]).max(axis=1) > 1e-4
    print(f"seed {seed}: {int(state.sum())} pixels depend on the texel state")
    assert np.abs(r["rgb"][state] - o["rgb"][state]).max(initial=0) < 2e-3
    assert np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
    assert sc.last_stats["rays_nearest"] == o["rays_nearest"] and sc.last_stats["rays_shadow"] == o["rays_shadow"]
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(o["rgb"]).astype(int)).max(axis=1)
    assert (d8 > 1).mean() < 1e-3, f"{(d8 > 1).sum()} pixels differ by more than 1/255"
    img = sc.render(W, H, camera=cam, max_depth=6, seed=7)
    assert np.abs(img.reshape(-1, 3) - r["rgb"]).max() < 1e-6
    sc.close()


def test_awkward_rays_on_a_random_scene(tmp_path):
    """rt_intersect_batch / rt_shadow_batch with directions that have zero, denormal, tiny or huge components (the slab
    test's 0 * inf cases), origins on box faces, lights at the shaded point (tools/fuzz_rays.py: 20 seeds agree)."""
    flat = rt.FlatScene.from_json(scenes.random_scene(str(tmp_path), seed=3))
    sc = rt.Scene(flat, 0)
    rng = np.random.default_rng(3)
    e, d = scenes.awkward_rays(rng, 60000)
    r, o = sc.intersect_batch(e, d), oracle.intersect_batch(flat, e, d)
    assert np.array_equal(r["hit"], o["hit"]) and np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
    hit = o["hit"].astype(bool)
    p = (e + o["t"][:, None] * d)[hit].astype(np.float32)
    l = rng.uniform([-4, 2, -9], [4, 6, 1], (len(p), 3)).astype(np.float32)
    l[::7] = p[::7]
    rs, os_ = sc.shadow_batch(p, l), oracle.shadow_batch(flat, p, l)
    assert np.array_equal(rs["lit"], os_["lit"])
    lit = os_["lit"].astype(bool)
    assert np.abs(rs["shadow"][lit] - os_["shadow"][lit]).max() < 1e-6
    sc.close()


def test_c4_area_lights_64_samples(tmp_path):
    flat = rt.FlatScene.from_json(scenes.c4_scene(str(tmp_path), samples=64))
    sc = rt.Scene(flat, 0)
    W, H = 240, 135
    cam = rt.Camera(flat.fov, W, H)
    s = primary_targets_np(cam, W, H)
    e = np.zeros((W * H, 3), np.float32)
    r = sc.trace_batch(e, s, seed=99)
    o = oracle.trace_batch(flat, e, s, seed=99, threads=8)
    assert np.array_equal(r["id"], o["id"])
    assert sc.last_stats["rays_shadow"] == o["rays_shadow"] == 128 * int(o["hit"].sum())
    assert np.abs(r["rgb"] - o["rgb"]).mean() < 1e-5 and np.abs(r["rgb"] - o["rgb"]).max() < 5e-3   # same sample points on both sides
    # the frame path (packets of 64 samples per surface) agrees with the batch path
    img = sc.render(W, H, camera=cam, seed=99)
    assert np.abs(img.reshape(-1, 3) - r["rgb"]).max() < 1e-6
    sc.close()


def test_c4_full_size_frame_properties(tmp_path):
    """C4 at BASELINE's size -- 1920x1080, two area lights of 64 samples -- through size-independent properties: the frame is the
    same bits run to run, the tiles of 2 and of 4 shards compose to it, and every hit costs exactly 128 shadow rays."""
    import torch
    flat = rt.FlatScene.from_json(scenes.c4_scene(str(tmp_path), samples=64))
    sc = rt.Scene(flat, 0)
    W, H = 1920, 1080
    cam = rt.Camera(flat.fov, W, H)
    a = torch.zeros(H, W, 3, device="cuda")
    st = sc.render_into(a.data_ptr(), W, H, camera=cam, seed=7)
    b = torch.zeros(H, W, 3, device="cuda")
    sc.render_into(b.data_ptr(), W, H, camera=cam, seed=7)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    s = primary_targets_np(cam, W, H)
    hits = int(sc.intersect_batch(np.zeros((W * H, 3), np.float32), s)["hit"].sum())
    assert st["rays_nearest"] == W * H and st["rays_shadow"] == 128 * hits and st["levels"] == 1
    for shards in (2, 4):
        parts = torch.zeros(H, W, 3, device="cuda")
        rays = 0
        for k in range(shards):
            rays += sc.render_into(parts.data_ptr(), W, H, camera=cam, seed=7, tile_rank=k, tile_ranks=shards)["rays_shadow"]
        assert torch.equal(parts.view(torch.int32), a.view(torch.int32)) and rays == st["rays_shadow"]
    c = torch.zeros(H, W, 3, device="cuda")
    sc.render_into(c.data_ptr(), W, H, camera=cam, seed=8)                 # another seed: other sample points, another image
    assert not torch.equal(a, c) and float((a - c).abs().mean()) < 0.02
    sc.close()


def test_scene_l_at_640x480_counts_the_hidden_state_pixels(gpu_scenes, flat_scenes):
    """Bundled scene l at the reference's own 640x480 against the compiled reference's dump.  Two pixels of the reference depend on the hit
    cache a mesh was left with by exit searches of EARLIER pixels (objects.cpp:123-125): the device, which has no
    state across pixels, may differ there and nowhere else."""
    import os
    from conftest import GOLD
    g = dict(np.load(os.path.join(GOLD, "ref_l_640.npz")))
    fs, sc = flat_scenes("l"), gpu_scenes("l")
    cam = rt.Camera(fs.fov, 640, 480)
    img = sc.render(640, 480, camera=cam)
    st = dict(sc.last_stats)
    got = rt.quantize_rgb8(img)[::-1].reshape(-1, 3)          # back to row 0 = bottom, the dump's order
    off = (np.abs(got.astype(int) - g["rgb8"].astype(int)).max(axis=1) > 1)
    print(f"scene l 640x480: {int(off.sum())} of 307200 pixels beyond 1/255 of the compiled reference (hidden-state pixels: at most 2)")
    assert int(off.sum()) <= 2
    assert abs(st["rays_nearest"] - int(g["rays_nearest"])) <= 4 and abs(st["rays_shadow"] - int(g["rays_shadow"])) <= 8
    # The library reports that class itself: rt_stats.state_cases counts the exit searches that found no exit (objects.cpp:123-125), the only
    # place where the reference reads state of earlier pixels.  The oracle counts the same event (oracle_state_cases) on the same ray trees.
    oracle.state_cases(reset=True)
    oracle.render(fs, cam, 640, 480, seed=1, threads=8)
    want_cases = oracle.state_cases(reset=True)
    print(f"scene l 640x480: {st['state_cases']} failed exit searches on the device, {want_cases} in the oracle")
    assert st["state_cases"] == want_cases and want_cases >= int(off.sum())
    s = primary_targets_np(cam, 640, 480)
    r = sc.intersect_batch(np.zeros((640 * 480, 3), np.float32), s)
    assert np.array_equal(r["id"].astype(np.int16), g["id"])


@pytest.mark.parametrize("name", list("acgijk"))
def test_bundled_scenes_at_640x480_vs_reference_dump(name, gpu_scenes, flat_scenes):
    """C1 at its stated size: the bundled scenes at the reference's own 640x480 against the compiled reference's dumps
    (tests/golden/ref_<scene>_640.npz: packed hit mask, ids as int16, 8-bit RGB, both ray counts).  Mask, ids and ray counts exact,
    every pixel of the 8-bit frame within 1."""
    import os
    from conftest import GOLD
    g = dict(np.load(os.path.join(GOLD, f"ref_{name}_640.npz")))
    fs, sc = flat_scenes(name), gpu_scenes(name)
    cam = rt.Camera(fs.fov, 640, 480)
    img = sc.render(640, 480, camera=cam)
    st = dict(sc.last_stats)
    got = rt.quantize_rgb8(img)[::-1].reshape(-1, 3)          # back to row 0 = bottom, the dump's order
    d8 = np.abs(got.astype(int) - g["rgb8"].astype(int)).max(axis=1)
    print(f"scene {name} 640x480: {int((d8 > 0).sum())} of 307200 pixels differ in the 8-bit frame, max {int(d8.max())}")
    assert int((d8 > 1).sum()) == 0
    assert st["rays_nearest"] == int(g["rays_nearest"]) and st["rays_shadow"] == int(g["rays_shadow"])
    s = primary_targets_np(cam, 640, 480)
    r = sc.intersect_batch(np.zeros((640 * 480, 3), np.float32), s)
    assert np.array_equal(r["id"].astype(np.int16), g["id"])
    assert np.array_equal(np.packbits(r["hit"] != 0), g["hit"])


def test_c5_full_size_4k_16spp_properties(tmp_path):
    """C5 at BASELINE's size -- 4 999 122 triangles, 3840x2160, 4x4 samples per pixel -- through size-independent properties: ray counts
    (16 primary rays per pixel, two shadow rays per hit), the same bits run to run, and the tiles of 8 shards compose to the frame."""
    import torch
    flat = rt.FlatScene.from_json(scenes.c5_scene(str(tmp_path), grid=1581))
    sc = rt.Scene(flat, 0)
    W, H = 3840, 2160
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    a = torch.zeros(H, W, 3, device="cuda")
    st = sc.render_into(a.data_ptr(), W, H, aa=(4, 4), camera=cam)
    assert st["rays_nearest"] == 16 * W * H and st["rays_shadow"] % 2 == 0 and 0 < st["rays_shadow"] <= 2 * st["rays_nearest"]
    parts = torch.zeros(H, W, 3, device="cuda")
    rays_n = rays_s = 0
    for k in range(8):
        p = sc.render_into(parts.data_ptr(), W, H, aa=(4, 4), camera=cam, tile_rank=k, tile_ranks=8)
        rays_n += p["rays_nearest"]; rays_s += p["rays_shadow"]
    assert torch.equal(parts.view(torch.int32), a.view(torch.int32))         # 8-shard composition, and with it run-to-run determinism
    assert rays_n == st["rays_nearest"] and rays_s == st["rays_shadow"]
    # the 16 samples of a pixel average to what 16 single-sample frames shifted by the sub-pixel offsets give: checked on the supersampling
    # identity of a coarse frame instead (cheap): a 4x4-sample frame at W/4 is the box filter of a 1-sample frame at W whose pixel centres
    # are the sub-pixel positions
    w4, h4 = W // 4, H // 4
    cam4 = rt.Camera(flat.fov, w4, h4, eye=(0, 1.5, 1.0), rot_x=-0.3)
    coarse = torch.zeros(h4, w4, 3, device="cuda")
    sc.render_into(coarse.data_ptr(), w4, h4, aa=(4, 4), camera=cam4)
    fine = torch.zeros(H, W, 3, device="cuda")
    sc.render_into(fine.data_ptr(), W, H, camera=cam)
    box = fine.view(h4, 4, w4, 4, 3).mean(dim=(1, 3))
    assert float((box - coarse).abs().max()) < 2e-6
    sc.close()


def test_c5_style_large_mesh_deep_tree(tmp_path):
    # 2*300^2 = 180 000 triangles: more than 2^17 objects, so the reference's depth cap (bvh.h:6) leaves several per leaf
    flat = rt.FlatScene.from_json(scenes.c5_scene(str(tmp_path), grid=300))
    sc = rt.Scene(flat, 0)
    W, H = 320, 180
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    s = primary_targets_np(cam, W, H)
    e = np.tile(np.array(cam.c.eye[:], np.float32), (W * H, 1))
    r = sc.trace_batch(e, s)
    o = oracle.trace_batch(flat, e, s, seed=1, threads=8)
    assert np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
    assert np.abs(rgb8(r["rgb"]).astype(int) - rgb8(o["rgb"]).astype(int)).max() <= 1
    img = sc.render(W, H, aa=(4, 4), camera=cam)       # 16 samples per pixel, the C5 sampling
    ref, nn, ns = oracle.render(flat, cam, W, H, aa=(4, 4), seed=1, threads=8)
    assert sc.last_stats["rays_nearest"] == nn and sc.last_stats["rays_shadow"] == ns
    assert np.abs(rgb8(img).astype(int) - rgb8(ref).astype(int)).max() <= 1
    sc.close()


def test_fat_leaves_of_64_triangles(tmp_path, monkeypatch):
    """4 193 408 triangles under the depth-capped tree are 65 536 leaves of 63 / 64: their sub-trees have nodes of 13 .. 16 triangles whose
    four grandchildren are sub-leaves of up to four -- more triangles than fit a treelet beside their quad (such a node keeps its two
    children as entries).  The packet walk against the per-lane walk on the same frame, bit for bit."""
    import torch
    flat = rt.FlatScene.from_json(scenes.c5_scene(str(tmp_path), grid=1448))
    assert flat.info.n_tris == 2 * 1448 * 1448
    sc = rt.Scene(flat, 0)
    W, H = 480, 270
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    a = sc.render(W, H, camera=cam)
    monkeypatch.setenv("RT_PACKET_LEVELS", "0")
    b = sc.render(W, H, camera=cam)
    assert np.array_equal(bits(a), bits(b))
    sc.close()


@pytest.mark.parametrize("textured", [False, True])
def test_deeper_levels_in_packets(tmp_path, monkeypatch, textured):
    """RT_PACKET_LEVELS > 1 sends the deeper levels of the ray tree through the packet kernels too (64 consecutive children of the level
    above per packet; not the default, see Run::packetLevels).  A mirror-like height field under a mirror sphere and a glass sphere:
    the frame with every level in packets against the oracle, and against the default frame and the all-per-lane frame bit for bit.
    Untextured: the level loop that runs ahead of the host (packet launches sized by a bound, counts read on the device); textured
    and reflective: the loop that reads the counts back after every level (texel records)."""
    import json
    src = scenes.c2_scene(str(tmp_path), grid=64)
    with open(src) as f:
        doc = json.load(f)
    mesh, s1, s2 = doc["objects"][:3]
    if not textured:
        mesh["type"] = "mesh"
        mesh.pop("texture"); mesh.pop("uvCoords")
        mesh["material"]["diffuse"] = [0.5, 0.6, 0.4]
    mesh["material"]["reflective"] = [0.4, 0.4, 0.4]
    for sph in (s1, s2):
        sph["type"] = "sphere"
        sph.pop("bumpmap"); sph.pop("bumpdepth")
    s1["material"]["reflective"] = [0.5, 0.5, 0.5]
    s2["material"]["transmissive"] = [0.6, 0.6, 0.6]
    s2["material"]["refraction"] = 1.3
    path = str(tmp_path / f"mirror_field_{int(textured)}.json")
    with open(path, "w") as f:
        json.dump(doc, f)
    flat = rt.FlatScene.from_json(path)
    sc = rt.Scene(flat, 0)
    W, H = 320, 180
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    monkeypatch.setenv("RT_PACKET_LEVELS", "8")
    a = sc.render(W, H, camera=cam, max_depth=6)
    st = dict(sc.last_stats)
    assert st["levels"] == 7
    ref, nn, ns = oracle.render(flat, cam, W, H, max_depth=6, seed=1, threads=8)
    assert st["rays_nearest"] == nn and st["rays_shadow"] == ns
    assert np.abs(rgb8(a).astype(int) - rgb8(ref).astype(int)).max() <= 1
    monkeypatch.delenv("RT_PACKET_LEVELS")
    b = sc.render(W, H, camera=cam, max_depth=6)
    assert sc.last_stats["rays_nearest"] == nn
    assert np.array_equal(bits(a), bits(b))
    monkeypatch.setenv("RT_PACKET_LEVELS", "0")
    c = sc.render(W, H, camera=cam, max_depth=6)
    assert np.array_equal(bits(a), bits(c))
    sc.close()


@pytest.mark.parametrize("aa", [(3, 1), (2, 2), (4, 4), (3, 5)])
def test_row_sums_of_multi_sample_frames_are_exact(tmp_path, monkeypatch, gpu_scenes, aa):
    """Frames with several samples per pixel add the fixed-point contributions of neighbouring lanes before the accumulator atomics
    (accumAddRows: the shading kernel and the two-light shadow kernel).  Integer sums, so the frame must equal, bit for bit, the one
    rendered with one atomic per contribution (RT_NO_ROW_SUMS): the bench scene at a small size (two point-type lights, runs of
    samples broken by misses), the mirror box of scene a (17 levels: children of one pixel scattered over the queues) and a frame
    whose width leaves partial tiles."""
    flat = rt.FlatScene.from_json(scenes.c2_scene(str(tmp_path), grid=48))
    sc = rt.Scene(flat, 0)
    cases = [(sc, 200, 120, dict(camera=rt.Camera(flat.fov, 200, 120, eye=(0, 1.5, 1.0), rot_x=-0.3))), (gpu_scenes("a"), 160, 120, {}), (gpu_scenes("h"), 75, 41, {})]
    for scene, W, H, kw in cases:
        monkeypatch.delenv("RT_NO_ROW_SUMS", raising=False)
        a = scene.render(W, H, aa=aa, **kw)
        rays = (scene.last_stats["rays_nearest"], scene.last_stats["rays_shadow"])
        monkeypatch.setenv("RT_NO_ROW_SUMS", "1")
        b = scene.render(W, H, aa=aa, **kw)
        assert rays == (scene.last_stats["rays_nearest"], scene.last_stats["rays_shadow"])
        assert np.array_equal(bits(a), bits(b))
    sc.close()


@pytest.mark.parametrize("name", ["a", "b", "c", "e", "g", "h", "i", "j", "k", "l", "m", "c3", "c4mirror", "c2mirror"])
def test_one_launch_per_deeper_level_equals_the_four_kernel_form(tmp_path, monkeypatch, gpu_scenes, name):
    """Below level 0 one launch walks, shades and lights a ray (k_level), and for small scenes -- all the reference ships -- with the scene's
    arrays copied into LDS first, and then level 0 goes the same way instead of through the packet kernels (RT_NO_SMALL_LEVEL0; c, j: bump
    maps and textures, k, m: one level only).  Same device functions and integer pixel sums, so the frame must equal, bit for bit and in both ray
    counts, the one the queues and the four kernels give (RT_NO_FUSED_LEVELS), the one walked in memory (RT_NO_SMALL_SCENE), the one with the
    exit searches of transmitted rays in k_transmit (RT_NO_INLINE_TRANSMIT) and the one whose counts k_publish reports (RT_NO_SELF_REPORT): mirrors
    (a), refraction (b, g: a refractive mesh, h: both at every hit), the 242-triangle meshes whose tree nearly fills the LDS budget (i),
    CSG (l, C3's 64 objects at depth 5), area lights, whose surface records k_level still writes (the C4 room with a mirror floor), and a
    scene too large for the LDS copy (a reflective 4 608-triangle height field)."""
    kw = {}
    if name == "c3":
        flat = rt.FlatScene.from_json(scenes.c3_scene(str(tmp_path)))
        sc, W, H = rt.Scene(flat, 0), 240, 135
        kw = dict(camera=rt.Camera(flat.fov, W, H, eye=(0, 1.0, 2.0), rot_x=-0.25), max_depth=5)
    elif name == "c4mirror":
        doc = json.load(open(scenes.c4_scene(str(tmp_path), 8)))
        plane = next(o for o in doc["objects"] if o["type"] == "plane")
        plane.setdefault("material", {})["reflective"] = [0.5, 0.5, 0.5]
        path = str(tmp_path / "c4mirror.json")
        json.dump(doc, open(path, "w"))
        flat = rt.FlatScene.from_json(path)
        sc, W, H = rt.Scene(flat, 0), 160, 90
        kw = dict(max_depth=4)
    elif name == "c2mirror":
        doc = json.load(open(scenes.c2_scene(str(tmp_path), grid=48)))
        mesh, s1, s2 = doc["objects"][:3]
        mesh["type"] = "mesh"
        mesh.pop("texture"); mesh.pop("uvCoords")
        mesh["material"]["diffuse"] = [0.5, 0.6, 0.4]
        mesh["material"]["reflective"] = [0.4, 0.4, 0.4]
        for sph in (s1, s2):
            sph["type"] = "sphere"
            sph.pop("bumpmap"); sph.pop("bumpdepth")
        s2["material"]["transmissive"] = [0.6, 0.6, 0.6]
        s2["material"]["refraction"] = 1.3
        path = str(tmp_path / "mirror.json")
        json.dump(doc, open(path, "w"))
        flat = rt.FlatScene.from_json(path)
        sc, W, H = rt.Scene(flat, 0), 200, 120
        kw = dict(camera=rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3), max_depth=6)
    else:
        sc, W, H = gpu_scenes(name), 160, 120
    switches = ("RT_NO_SMALL_LEVEL0", "RT_NO_SMALL_SCENE", "RT_NO_INLINE_TRANSMIT", "RT_NO_SELF_REPORT", "RT_NO_FUSED_LEVELS")
    for var in switches:
        monkeypatch.delenv(var, raising=False)
    a = sc.render(W, H, seed=5, **kw)
    st = dict(sc.last_stats)
    assert st["levels"] > 1 or name in "cjkm"
    for var in switches:
        monkeypatch.setenv(var, "1")
        b = sc.render(W, H, seed=5, **kw)
        assert (st["rays_nearest"], st["rays_shadow"], st["levels"]) == (sc.last_stats["rays_nearest"], sc.last_stats["rays_shadow"], sc.last_stats["levels"]), var
        assert np.array_equal(bits(a), bits(b)), var
        monkeypatch.delenv(var)
    if name in ("c3", "c2mirror", "c4mirror"):
        sc.close()


@pytest.mark.parametrize("seed,n_objects,analytic", [(3, 4, False), (6, 14, False), (4, 4, True), (5, 8, True), (14, 14, True), (7, 40, True)])
def test_one_launch_levels_on_random_scenes_vs_oracle(tmp_path, monkeypatch, seed, n_objects, analytic):
    """k_level on random scenes of every primitive, material and light kind (tools/fuzz_levels.py: the textured meshes non-reflective, so
    that no texel records switch it off; `analytic`: no CSG objects and opaque meshes, so exit searches run inline and the kernel reports
    its own counts): the frame against the oracle -- both ray counts equal, 8-bit RGB within 1 -- and against the four-kernel form."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_levels", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_levels.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    path, _ = fz.scene_for(str(tmp_path), seed, n_objects, analytic)
    flat = rt.FlatScene.from_json(path)
    sc = rt.Scene(flat, 0)
    W, H = 160, 90
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.0, 1.5), rot_x=-0.2)
    monkeypatch.delenv("RT_NO_FUSED_LEVELS", raising=False)
    img = sc.render(W, H, camera=cam, max_depth=6, seed=7)
    st = dict(sc.last_stats)
    ref, nn, ns = oracle.render(flat, cam, W, H, max_depth=6, seed=7, threads=8)
    assert (st["rays_nearest"], st["rays_shadow"]) == (nn, ns)
    assert np.abs(rgb8(img).astype(int) - rgb8(ref).astype(int)).max() <= 1
    monkeypatch.setenv("RT_NO_FUSED_LEVELS", "1")
    old = sc.render(W, H, camera=cam, max_depth=6, seed=7)
    assert (sc.last_stats["rays_nearest"], sc.last_stats["rays_shadow"]) == (nn, ns)
    assert np.abs(img - old).max() < 1e-6      # (bit-identical but for the order in which a shadow ray multiplies three or more transmissive occluders)
    sc.close()


@pytest.mark.parametrize("samples", [64, 24])
def test_deferred_csg_leaves_of_area_light_shadow_rays(tmp_path, monkeypatch, samples):
    """The area-light kernel notes the CSG objects its shadow rays reach and evaluates them when the walk is over (scenes with at most
    48 CSG objects; packetTraverse, DEFER).  The C4 room (two CSG objects) with 64 samples per light -- a wave is one surface -- and with
    24: the frame against the one with csgEval inside the walk (RT_NO_CSG_DEFER), bit for bit, and against the oracle's ray counts."""
    flat = rt.FlatScene.from_json(scenes.c4_scene(str(tmp_path), samples))
    sc = rt.Scene(flat, 0)
    W, H = 320, 180
    a = sc.render(W, H, seed=3)
    rays = (sc.last_stats["rays_nearest"], sc.last_stats["rays_shadow"])
    monkeypatch.setenv("RT_NO_CSG_DEFER", "1")
    b = sc.render(W, H, seed=3)
    assert rays == (sc.last_stats["rays_nearest"], sc.last_stats["rays_shadow"])
    assert np.array_equal(bits(a), bits(b))
    _, nn, ns = oracle.render(flat, rt.Camera(flat.fov, W, H), W, H, seed=3, threads=8)
    assert rays == (nn, ns)
    sc.close()


def test_c5_full_size_from_a_mesh_file(tmp_path):
    # 4 999 122 triangles through the binary mesh ingest: 65 536 leaves of ~76 triangles (depth cap), which the packet
    # kernels read as long external runs, 16 at a time
    flat = rt.FlatScene.from_json(scenes.c5_scene(str(tmp_path), grid=1581))
    assert flat.info.n_tris == 4999122
    sc = rt.Scene(flat, 0)
    W, H = 192, 108
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    img = sc.render(W, H, camera=cam)                                  # packets
    ref, nn, ns = oracle.render(flat, cam, W, H, seed=1, threads=8)
    assert sc.last_stats["rays_nearest"] == nn and sc.last_stats["rays_shadow"] == ns
    assert np.abs(rgb8(img).astype(int) - rgb8(ref).astype(int)).max() <= 1
    # sampled rays of the 4K frame: ids exact, t bit-exact (per-lane walk) ...
    W4, H4 = 3840, 2160
    cam4 = rt.Camera(flat.fov, W4, H4, eye=(0, 1.5, 1.0), rot_x=-0.3)
    s = primary_targets_np(cam4, W4, H4)
    pick = np.random.default_rng(11).choice(W4 * H4, 20000, replace=False)
    # ... among them pixel (1890, 1309): its ray hits the edge between triangles 2419408 and 2422569 with the same t bit for bit, and the
    # box of the one the reference finds first is entered an ulp beyond that t -- a walk that meets the other triangle first used to
    # prune it (round 2: boxes within the margin of the best hit are visited, the keys settle the tie) ...
    tie = 1309 * W4 + 1890
    pick = np.unique(np.concatenate([pick, [tie]]))
    e = np.tile(np.array(cam4.c.eye[:], np.float32), (len(pick), 1))
    r = sc.trace_batch(e, s[pick])
    o = oracle.trace_batch(flat, e, s[pick], seed=1, threads=8)
    assert np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
    assert o["id"][np.searchsorted(pick, tie)] == 2419408
    # ... and the 4K frame itself (packets) shows the same colours at those pixels
    frame = sc.render(W4, H4, camera=cam4).reshape(-1, 3)[pick]
    assert np.abs(rgb8(frame).astype(int) - rgb8(o["rgb"]).astype(int)).max() <= 1
    sc.close()


@pytest.mark.parametrize("kind", ["c2", "mixed", "csg", "fat_leaves"])
def test_packet_walks_on_caller_rays_vs_oracle(tmp_path, monkeypatch, kind):
    """The packet kernels (quad nodes, lane masks, one stack per wave) on rays they are never given in a frame: with RT_PACKET_BATCHES
    caller-supplied rays are walked as packets of 64 consecutive rays -- camera rays in scanline order (a packet is a 64 x 1 strip,
    not an 8 x 8 block), the same rays shuffled (64 unrelated rays per packet) -- and must give what the per-lane walk and the oracle
    give: ids exact, t bit for bit, colours (whose shadow rays go through the shadow packets) within 1/255."""
    if kind == "c2":
        flat, eye, rot = rt.FlatScene.from_json(scenes.c2_scene(str(tmp_path), grid=48)), (0.0, 1.5, 1.0), -0.3
    elif kind == "mixed":
        flat, eye, rot = rt.FlatScene.from_json(scenes.random_scene(str(tmp_path), seed=4)), (0.0, 0.0, 0.0), 0.0
    elif kind == "csg":
        flat, eye, rot = rt.FlatScene.from_json(scenes.random_csg_scene(str(tmp_path), seed=2, max_leaves=16)), (0.0, 1.0, 1.5), -0.2
    else:
        flat, eye, rot = rt.FlatScene.from_json(scenes.c5_scene(str(tmp_path), grid=300)), (0.0, 1.5, 1.0), -0.3
    sc = rt.Scene(flat, 0)
    W, H = 256, 144
    cam = rt.Camera(flat.fov, W, H, eye=eye, rot_x=rot)
    s = primary_targets_np(cam, W, H)
    e = np.tile(np.array(cam.c.eye[:], np.float32), (W * H, 1))
    o = oracle.trace_batch(flat, e, s, max_depth=3, seed=1, threads=8)
    perm = np.random.default_rng(5).permutation(W * H)
    monkeypatch.setenv("RT_PACKET_BATCHES", "1")
    for order in (np.arange(W * H), perm):
        r = sc.trace_batch(e[order], s[order], max_depth=3)
        assert np.array_equal(r["id"], o["id"][order]) and np.array_equal(bits(r["t"]), bits(o["t"][order]))
        assert sc.last_stats["rays_nearest"] == o["rays_nearest"] and sc.last_stats["rays_shadow"] == o["rays_shadow"]
        d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(o["rgb"][order]).astype(int)).max(axis=1)
        assert (d8 > 1).mean() < 1e-3, f"{(d8 > 1).sum()} pixels differ by more than 1/255"
    sc.close()


def test_meshes_asked_for_hits_on_their_own(tmp_path):
    # Mesh::rayhit through the per-mesh tree: a refractive 1280-triangle icosphere (exit search of transmitRay, internal
    # reflections) and a CSG difference with a 320-triangle mesh operand, against the oracle's linear scan
    flat = rt.FlatScene.from_json(scenes.mesh_scene(str(tmp_path), subdiv=3))
    assert flat.info.n_mesh_tris == 1280 + 320
    sc = rt.Scene(flat, 0)
    W, H = 320, 180
    cam = rt.Camera(flat.fov, W, H)
    s = primary_targets_np(cam, W, H)
    e = np.zeros((W * H, 3), np.float32)
    r = sc.trace_batch(e, s)
    o = oracle.trace_batch(flat, e, s, seed=1, threads=8)
    assert np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
    assert sc.last_stats["rays_nearest"] == o["rays_nearest"] and sc.last_stats["rays_shadow"] == o["rays_shadow"]
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(o["rgb"]).astype(int)).max(axis=1)
    assert (d8 > 1).mean() < 1e-3, f"{(d8 > 1).sum()} pixels differ by more than 1/255"
    assert len(np.unique(r["id"][r["hit"] != 0])) >= 4          # glass, carved mesh, floor, sphere all seen
    img = sc.render(W, H, camera=cam)
    assert np.abs(rgb8(img.reshape(-1, 3)).astype(int) - rgb8(o["rgb"]).astype(int)).max(axis=1).mean() < 0.01
    sc.close()


def test_light_pairs_and_single_light_packets_agree(tmp_path, monkeypatch):
    """Shadow packets with two lights per surface (two rays per lane) against one light per packet and against the oracle:
    four point-type lights (two pairs), then three (no pairing), on a scene with transmissive occluders."""
    import json
    path = scenes.random_scene(str(tmp_path), seed=38)
    scene = json.load(open(path))
    scene["lights"] = [l for l in scene["lights"] if l["type"] in ("ambient", "directional", "point")]
    scene["lights"] += [{"type": "point", "color": [0.3, 0.5, 0.4], "position": [-2.5, 3.5, -9.0]},
                        {"type": "spot", "color": [0.6, 0.5, 0.3], "position": [2.0, 4.0, -1.0], "direction": [-0.2, -1.0, -0.6], "cutoff": 40}]
    for n_lights in (5, 4):                                 # ambient + 4 / + 3 point-type lights
        scene["lights"] = scene["lights"][:n_lights]
        p = str(tmp_path / f"lights{n_lights}.json")
        json.dump(scene, open(p, "w"))
        flat = rt.FlatScene.from_json(p)
        sc = rt.Scene(flat, 0)
        W, H = 256, 144
        cam = rt.Camera(flat.fov, W, H, eye=(0, 1.0, 1.5), rot_x=-0.2)
        paired = sc.render(W, H, camera=cam, max_depth=4)
        st = dict(sc.last_stats)
        monkeypatch.setenv("RT_NO_LIGHT_PAIRS", "1")
        single = sc.render(W, H, camera=cam, max_depth=4)
        monkeypatch.delenv("RT_NO_LIGHT_PAIRS")
        assert np.abs(paired - single).max() < 1e-6 and sc.last_stats["rays_shadow"] == st["rays_shadow"]
        ref, nn, ns = oracle.render(flat, cam, W, H, max_depth=4, seed=1, threads=8)
        assert st["rays_nearest"] == nn and st["rays_shadow"] == ns
        assert (np.abs(rgb8(paired).astype(int) - rgb8(ref).astype(int)).max(axis=2) > 1).mean() < 1e-3
        sc.close()


@pytest.mark.parametrize("queue_rays,W,H,aa", [(None, 640, 360, 1), (32768, 640, 360, 1), (None, 320, 180, 2)])
def test_texel_state_with_many_records(tmp_path, monkeypatch, queue_rays, W, H, aa):
    """A reflective textured terrain under a mirror: nearly every ray tree comes back to a textured triangle it has hit before, so
    tens of thousands of hits are lit with a later hit's texel (texturemesh.cpp:49-50 / objects.cpp:62-79; k_texel_fix).  On a frame
    cut into four shards, with ray queues (and with them the record buffer) so small that batches overflow and are re-run, and with
    four samples per pixel (the records of a pixel's samples share one list)."""
    import json
    from raytracer_amd.scenes import _mesh_json, height_field, tiles_texture, write_bmp24
    write_bmp24(str(tmp_path / "t.bmp"), tiles_texture(64, 8))
    tris, uvs = height_field(6, seed=3, x0=-6.0, x1=6.0, z0=-14.0, z1=-1.0)   # large triangles: rays come back to the same one
    mesh = {"type": "texturemesh", "texture": "t.bmp", "material": {"specular": [0.3, 0.3, 0.3], "shininess": 10.0, "reflective": [0.5, 0.4, 0.6]}}
    mesh.update(_mesh_json(tris, uvs))
    scene = {"camera": {"field": 60, "background": [0.1, 0.1, 0.2]},
             "objects": [mesh, {"type": "box", "point1": [-8, 1.5, -16], "point2": [8, 2.0, 1], "material": {"diffuse": [0.1, 0.1, 0.1], "reflective": [0.8, 0.8, 0.8]}},
                         {"type": "sphere", "position": [0.5, 0.2, -6], "radius": 0.7, "material": {"diffuse": [0.5, 0.2, 0.2], "transmissive": [0.6, 0.6, 0.6], "refraction": 1.3}}],
             "lights": [{"type": "ambient", "color": [0.3, 0.3, 0.3]}, {"type": "point", "color": [0.8, 0.8, 0.7], "position": [0, 1.2, -5]},
                        {"type": "directional", "color": [0.3, 0.3, 0.4], "direction": [0.2, -1, -0.3]},
                        {"type": "rectangular", "color": [0.4, 0.4, 0.3], "position": [1.0, 1.4, -7.0], "normal": [0, -1, 0], "orientation": [1, 0, 0],
                         "width": 1.0, "height": 0.6, "samples": 3}]}
    path = tmp_path / "mirror_terrain.json"
    path.write_text(json.dumps(scene))
    flat = rt.FlatScene.from_json(str(path))
    if queue_rays:
        monkeypatch.setenv("RT_QUEUE_RAYS", str(queue_rays))
    sc = rt.Scene(flat, 0)
    depth = 5
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.0, -7.0), rot_x=-1.3)   # between the two, looking down: the bounces stay on one triangle
    img = sc.render(W, H, aa=aa, camera=cam, max_depth=depth, seed=9)
    splits = sc.last_stats["batch_splits"]
    nn, ns = sc.last_stats["rays_nearest"], sc.last_stats["rays_shadow"]
    # the frame above went through the level loop that runs ahead of the host (the records' count comes with k_publish's reports);
    # with per-kernel timing asked for, the loop that reads the counters back after every level runs: the same frame, bit for bit
    timed = sc.render(W, H, aa=aa, camera=cam, max_depth=depth, seed=9, flags=2)
    assert (sc.last_stats["rays_nearest"], sc.last_stats["rays_shadow"]) == (nn, ns)
    assert np.array_equal(bits(img), bits(timed))
    sc.close()
    oracle.texel_cases()
    ref, onn, ons = oracle.render(flat, cam, W, H, aa=aa, max_depth=depth, seed=9, threads=8)
    cases = oracle.texel_cases()
    with oracle.own_texels():
        own, _, _ = oracle.render(flat, cam, W, H, aa=aa, max_depth=depth, seed=9, threads=8)
    state = np.abs(own - ref).max(axis=2) > 1e-3
    print(f"{cases} hits lit with a later hit's texel, {int(state.sum())} of {W * H} pixels visibly; {splits} batch splits")
    assert cases > 10000 and state.sum() > 2000
    assert (nn, ns) == (onn, ons)
    assert np.abs(img - ref).max() < 2e-4
    assert bool(splits) == bool(queue_rays)
