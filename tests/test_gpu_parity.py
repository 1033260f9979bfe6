"""GPU suite: the HIP path through the C-ABI against (1) dumps of the real reference (tests/golden) and (2) the oracle
restatement run on the same inputs on this machine.  Bars (north_star): hit masks and object ids exact, t bit-exact,
deterministic RGB within 1/255 on the 8-bit image, stochastic area lights within a stated mean-error tolerance."""
import numpy as np
import pytest

import oracle
from conftest import SCENES_DET, SCENES_RANDOM

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rgb8(rgb):
    q = np.clip(np.nan_to_num(np.asarray(rgb, np.float32), nan=0.0), 0, 1)
    return (q * np.float32(255) + np.float32(0.5)).astype(np.uint8)


@pytest.mark.parametrize("name", list(SCENES_DET))
def test_trace_batch_vs_reference_dump(name, gpu_scenes, golden):
    g = golden(name)
    n = int(g["W"]) * int(g["H"])
    sc = gpu_scenes(name)
    r = sc.trace_batch(np.zeros((n, 3), np.float32), g["s"])
    assert np.array_equal(r["hit"], g["hit"]), f"hit mask differs in {(r['hit'] != g['hit']).sum()} of {n} rays"
    assert np.array_equal(r["id"], g["id"]), f"object id differs in {(r['id'] != g['id']).sum()} of {n} rays"
    assert np.array_equal(bits(r["t"]), bits(g["t"])), f"t differs in {(bits(r['t']) != bits(g['t'])).sum()} rays"
    # RGB: within 1/255 on the 8-bit image; report how many pixels differ at all
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(g["rgb"]).astype(int))
    assert d8.max() <= 1, f"max 8-bit difference {d8.max()}"
    assert (d8 > 0).any(axis=1).mean() < 0.02
    assert np.nanmax(np.abs(r["rgb"] - g["rgb"])) < 2e-3
    # the ray counts of the frame equal the reference's BVH::findNearest / BVH::calcShadow call counts
    assert sc.last_stats["rays_nearest"] == int(g["rays_nearest"])
    assert sc.last_stats["rays_shadow"] == int(g["rays_shadow"])


@pytest.mark.parametrize("name", list(SCENES_DET) + list(SCENES_RANDOM))
def test_secondary_and_shadow_batches_vs_reference_dump(name, gpu_scenes, golden):
    g = golden(name)
    sc = gpu_scenes(name)
    if len(g["sec_id"]):
        r = sc.intersect_batch(g["sec_e"], g["sec_d"])
        assert np.array_equal(r["id"], g["sec_id"])
        assert np.array_equal(bits(r["t"]), bits(g["sec_t"]))
    r = sc.shadow_batch(g["shd_p"], g["shd_l"])
    assert np.array_equal(r["lit"], g["shd_lit"].astype(np.uint8))
    lit = r["lit"] != 0
    # the attenuation is a product over transmissive occluders; the visiting order may differ, so allow rounding
    assert np.allclose(r["shadow"][lit], g["shd_shadow"][lit], rtol=1e-6, atol=0)


@pytest.mark.parametrize("name", list(SCENES_RANDOM))
def test_random_scene_vs_reference_dump(name, gpu_scenes, flat_scenes, golden):
    """The compiled reference on a random scene holding every primitive, material and light kind.  Ids, t, ray counts and RGB as on
    the bundled scenes.  random_5 has two pixels where the reference lights a textured triangle with the texel of a LATER hit on it
    (texturemesh.cpp:49-50 / objects.cpp:62-79, k_texel_fix on the device): the oracle with that state switched off shows which."""
    g = golden(name)
    n = int(g["W"]) * int(g["H"])
    e = np.zeros((n, 3), np.float32)
    sc = gpu_scenes(name)
    r = sc.trace_batch(e, g["s"])
    assert np.array_equal(r["hit"], g["hit"]) and np.array_equal(r["id"], g["id"]) and np.array_equal(bits(r["t"]), bits(g["t"]))
    assert sc.last_stats["rays_nearest"] == int(g["rays_nearest"]) and sc.last_stats["rays_shadow"] == int(g["rays_shadow"])
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(g["rgb"]).astype(int)).max(axis=1)
    assert d8.max() <= 1
    assert np.nanmax(np.abs(r["rgb"] - g["rgb"])) < 2e-3
    with oracle.own_texels():
        own = oracle.trace_batch(flat_scenes(name), e, g["s"], seed=1, threads=4)
    state = np.abs(own["rgb"] - g["rgb"]).max(axis=1) > 1e-4
    assert state.sum() == {"random_5": 2, "random_8": 0}[name]
    assert np.abs(r["rgb"][state] - g["rgb"][state]).max(initial=0) < 1e-5      # ... and the device follows the reference there
    # the frame path (lanes, shards) gives the same pixels, and so does the picked ray (trace(..., pick = true))
    img = sc.render(int(g["W"]), int(g["H"]))
    assert np.abs(img.reshape(-1, 3) - r["rgb"]).max() < 1e-6
    for i in np.where(state)[0]:
        hit, rgb, text = sc.trace_pick(e[i], g["s"][i])
        assert hit and np.abs(np.asarray(rgb) - g["rgb"][i]).max() < 1e-5 and text.count("object triangle hit at") >= 3


def test_c2_family_vs_reference_dump(gpu_scenes, golden):
    """The benchmark's workload family against the COMPILED REFERENCE: the C2 generator with 4 608 triangles at 160x90 from bench.py's
    camera (tests/golden/ref_c2_48.npz).  Ids, t, ray counts exact; RGB within 1/255 on every pixel, the bump-mapped spheres included
    (their texel index is int(u*W) with u from atan2f / asinf, bump.cpp:14-23: csrc/glibc_math.h restates glibc's versions)."""
    g = golden("c2_48")
    n = int(g["W"]) * int(g["H"])
    sc = gpu_scenes("c2_48")
    e = np.tile(g["eye"].astype(np.float32), (n, 1))
    r = sc.trace_batch(e, g["s"])
    assert np.array_equal(r["hit"], g["hit"]) and np.array_equal(r["id"], g["id"]) and np.array_equal(bits(r["t"]), bits(g["t"]))
    assert sc.last_stats["rays_nearest"] == int(g["rays_nearest"]) and sc.last_stats["rays_shadow"] == int(g["rays_shadow"])
    d8 = np.abs(rgb8(r["rgb"]).astype(int) - rgb8(g["rgb"]).astype(int)).max(axis=1)
    print(f"{int((d8 > 1).sum())} of {n} pixels beyond 1/255, {int((d8 > 0).sum())} differ at all")
    assert (d8 > 1).sum() == 0
    a = sc.shadow_batch(g["shd_p"], g["shd_l"])
    assert np.array_equal(a["lit"], g["shd_lit"].astype(np.uint8))
    lit = a["lit"] != 0
    assert np.allclose(a["shadow"][lit], g["shd_shadow"][lit], rtol=1e-6, atol=0)


def test_awkward_rays_vs_reference_dump(gpu_scenes):
    """rt_intersect_batch / rt_shadow_batch against the COMPILED REFERENCE's BVH::findNearest / BVH::calcShadow for 20 000 awkward rays on
    the random_5 scene (zero, denormal, tiny and huge direction components, no direction at all, origins on box faces) and 8 820 shadow
    queries from their hit points (tests/golden/ref_rays_random_5.npz): ids and lit flags exact, t bit for bit."""
    import os
    from conftest import GOLD
    g = np.load(os.path.join(GOLD, "ref_rays_random_5.npz"))
    sc = gpu_scenes("random_5")
    r = sc.intersect_batch(g["e"], g["d"])
    hit = g["id"] >= 0
    assert np.array_equal(r["hit"] != 0, hit) and np.array_equal(r["id"][hit], g["id"][hit]) and np.array_equal(bits(r["t"][hit]), bits(g["t"][hit]))
    a = sc.shadow_batch(g["p"], g["l"])
    lit = g["lit"] != 0
    assert np.array_equal(a["lit"], g["lit"])
    assert np.allclose(a["shadow"][lit], g["shadow"][lit], rtol=1e-6, atol=0)


def test_scene_m_same_samples_as_oracle(gpu_scenes, flat_scenes, golden):
    # device and oracle share the counter-based generator, so with equal seeds they shade the same light samples
    g = golden("m")
    n = int(g["W"]) * int(g["H"])
    e = np.zeros((n, 3), np.float32)
    sc = gpu_scenes("m")
    r = sc.trace_batch(e, g["s"], seed=777)
    o = oracle.trace_batch(flat_scenes("m"), e, g["s"], seed=777, threads=8)
    assert np.array_equal(r["id"], o["id"]) and np.array_equal(bits(r["t"]), bits(o["t"]))
    assert np.abs(r["rgb"] - o["rgb"]).max() < 5e-3          # a sample flipping lit/shadowed moves a pixel by <= 1/50 of one light
    assert np.abs(r["rgb"] - o["rgb"]).mean() < 1e-5
    assert sc.last_stats["rays_shadow"] == o["rays_shadow"] == 19200 * 100


def test_scene_m_vs_high_sample_reference(gpu_scenes, golden):
    import os
    from conftest import GOLD
    g, hi = golden("m"), dict(np.load(os.path.join(GOLD, "ref_m_hi.npz")))
    n = int(g["W"]) * int(g["H"])
    r = gpu_scenes("m").trace_batch(np.zeros((n, 3), np.float32), g["s"], seed=4242)
    err = np.abs(np.clip(r["rgb"], 0, 1) - np.clip(hi["rgb"], 0, 1)).mean()
    err_ref = np.abs(np.clip(g["rgb"], 0, 1) - np.clip(hi["rgb"], 0, 1)).mean()
    assert err < 0.0125 and err < 1.1 * err_ref      # stated tolerance, see test_oracle_golden.py
    assert abs(float(r["rgb"].mean()) - float(hi["rgb"].mean())) < 1e-3


@pytest.mark.parametrize("name,aa", [("c", 1), ("c", 2), ("a", 1), ("k", 1), ("j", 1), ("h", 1)])
def test_render_frame_vs_oracle(name, aa, gpu_scenes, flat_scenes):
    import raytracer_amd as rt
    fs = flat_scenes(name)
    W, H = 160, 120
    cam = rt.Camera(fs.fov, W, H)
    sc = gpu_scenes(name)
    img = sc.render(W, H, aa=aa, camera=cam)
    ref, nn, ns = oracle.render(fs, cam, W, H, aa=aa, seed=1, threads=8)
    d8 = np.abs(rgb8(img).astype(int) - rgb8(ref).astype(int))
    assert d8.max() <= 1
    assert sc.last_stats["rays_nearest"] == nn and sc.last_stats["rays_shadow"] == ns


def test_frames_from_moved_and_turned_cameras(gpu_scenes):
    """k_generate against the compiled reference's primary rays (tests/golden/ref_cameras.npz: setFacing() and s(x, y), q1.cpp:30-53, for
    twenty cameras): the frame path gives, bit for bit, what rt_trace_batch gives for the reference's own eye and targets."""
    import os
    import raytracer_amd as rt
    from conftest import GOLD
    g = np.load(os.path.join(GOLD, "ref_cameras.npz"))
    sc = gpu_scenes("h")
    for k, c in enumerate(g["cameras"]):
        for W, H in ((16, 9), (7, 13)):
            cam = rt.Camera(float(g["fov"]), W, H, eye=tuple(c[:3]), rot_x=float(c[3]), rot_y=float(c[4]))
            img = sc.render(W, H, camera=cam)
            r = sc.trace_batch(np.tile(g[f"eye_{k}_{W}x{H}"], (W * H, 1)), g[f"s_{k}_{W}x{H}"])
            assert np.array_equal(bits(img.reshape(-1, 3)), bits(r["rgb"])), (k, W, H)


def test_render_is_deterministic_and_tiles_compose(gpu_scenes, flat_scenes):
    import torch
    import raytracer_amd as rt
    fs = flat_scenes("h")
    W, H = 200, 100          # not a multiple of the 32x8 tile: partial edge tiles
    cam = rt.Camera(fs.fov, W, H)
    sc = gpu_scenes("h")
    a = sc.render(W, H, camera=cam)
    b = sc.render(W, H, camera=cam)
    assert np.array_equal(bits(a), bits(b))          # fixed-point accumulation: bit-identical run to run
    full = torch.zeros(H, W, 3, device="cuda")
    for rank in range(3):
        sc.render_into(full.data_ptr(), W, H, camera=cam, tile_rank=rank, tile_ranks=3)
    assert np.array_equal(bits(full.cpu().numpy()), bits(a))


def test_area_light_frames_are_deterministic_whatever_the_queue_order(flat_scenes, monkeypatch):
    """Scene m (one area light, 100 samples: a surface's samples straddle waves, and where depends on the order in which the shading
    kernel's workgroups appended their surfaces): per-sample fixed-point shares added as integers, so the frame is the same bit for
    bit run to run, on one lane or four, for the whole frame or three ranks' tiles, and through the level loop that reads its
    counters back (flags = 2: per-kernel timing)."""
    import torch
    import raytracer_amd as rt
    fs = flat_scenes("m")
    W, H = 200, 100
    cam = rt.Camera(fs.fov, W, H)
    sc = rt.Scene(fs, 0)
    a = sc.render(W, H, camera=cam, seed=5)
    for _ in range(3):
        assert np.array_equal(bits(a), bits(sc.render(W, H, camera=cam, seed=5)))
    assert np.array_equal(bits(a), bits(sc.render(W, H, camera=cam, seed=5, flags=2)))
    full = torch.zeros(H, W, 3, device="cuda")
    for rank in range(3):
        sc.render_into(full.data_ptr(), W, H, camera=cam, seed=5, tile_rank=rank, tile_ranks=3)
    assert np.array_equal(bits(full.cpu().numpy()), bits(a))
    sc.close()
    monkeypatch.setenv("RT_LANES", "1")
    one = rt.Scene(fs, 0)
    assert np.array_equal(bits(a), bits(one.render(W, H, camera=cam, seed=5)))
    one.close()


def test_tiles_exchange_with_8_bit_pixels_equals_host_quantiser(gpu_scenes, flat_scenes):
    """The exchange with 8-bit pixels (rt_tiles_pack_rgb8 / _unpack_rgb8 / _quantize_rgb8): the frame assembled from three ranks'
    byte payloads is, byte for byte, rt_quantize_rgb8 of the single-rank float frame -- NaN, negative and > 1 pixels included."""
    import torch
    import raytracer_amd as rt
    fs = flat_scenes("c")
    W, H, ranks = 200, 100, 3          # partial edge tiles, ragged tile counts
    cam = rt.Camera(fs.fov, W, H)
    sc = gpu_scenes("c")
    whole = torch.from_numpy(sc.render(W, H, camera=cam)).cuda()
    whole[3, 5] = torch.tensor([float("nan"), -0.25, 7.0], device="cuda")      # the quantiser's corner cases
    whole[H - 1, W - 1] = torch.tensor([1.0, 0.0, 0.5], device="cuda")
    want = rt.quantize_rgb8(whole.cpu().numpy())
    frame8 = torch.full((H, W, 3), 99, dtype=torch.uint8, device="cuda")
    sc.tiles_quantize_rgb8(whole.data_ptr(), frame8.data_ptr(), W, H, 0, ranks)      # the root's own share
    for rank in range(1, ranks):
        n = sc.tiles_count(W, H, rank, ranks)
        packed = torch.full((n * 256 * 3,), 77, dtype=torch.uint8, device="cuda")
        sc.tiles_pack_rgb8(whole.data_ptr(), packed.data_ptr(), W, H, rank, ranks)
        sc.tiles_unpack_rgb8(packed.data_ptr(), frame8.data_ptr(), W, H, rank, ranks)
    torch.cuda.synchronize()
    assert np.array_equal(frame8.cpu().numpy(), want)
    # (0, 1) quantises the whole frame in one call
    all8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    sc.tiles_quantize_rgb8(whole.data_ptr(), all8.data_ptr(), W, H, 0, 1)
    torch.cuda.synchronize()
    assert np.array_equal(all8.cpu().numpy(), want)


def test_tiles_pack_gather_unpack_on_device(gpu_scenes, flat_scenes):
    """The multi-GPU assembly (bench.py, raytracer_amd.distributed) on one device: every rank's tiles packed, unpacked into
    the root's frame -- stream-ordered calls, no host synchronisation in between -- give the single-rank frame."""
    import torch
    import raytracer_amd as rt
    from raytracer_amd import distributed as rd
    fs = flat_scenes("c")
    W, H, ranks = 200, 100, 3          # partial edge tiles, ragged tile counts
    cam = rt.Camera(fs.fov, W, H)
    sc = gpu_scenes("c")
    whole = sc.render(W, H, camera=cam)
    root = torch.zeros(H, W, 3, device="cuda")
    for rank in range(ranks):
        part = torch.full((H, W, 3), -1.0, device="cuda")
        sc.render_into(part.data_ptr(), W, H, camera=cam, tile_rank=rank, tile_ranks=ranks)
        n = sc.tiles_count(W, H, rank, ranks)
        assert n == rd.tiles_count(W, H, rank, ranks)
        packed = torch.zeros(n * 256 * 3, device="cuda")
        sc.tiles_pack(part.data_ptr(), packed.data_ptr(), W, H, rank, ranks)
        sc.tiles_unpack(packed.data_ptr(), root.data_ptr(), W, H, rank, ranks)
        # the device's pack order is the one the CPU restatement (gloo tests) uses
        ref = rd.pack_tiles(part.cpu(), rank, ranks)
        assert np.array_equal(bits(packed.cpu().numpy().reshape(-1, 3)), bits(ref.numpy()))
    assert np.array_equal(bits(root.cpu().numpy()), bits(whole))


def test_queue_overflow_splits_batches(flat_scenes, monkeypatch):
    import raytracer_amd as rt
    monkeypatch.setenv("RT_QUEUE_RAYS", "4096")
    fs = flat_scenes("h")     # reflective AND transmissive sphere: the ray tree forks
    small = rt.Scene(fs, 0)
    monkeypatch.delenv("RT_QUEUE_RAYS")
    big = rt.Scene(fs, 0)
    cam = rt.Camera(fs.fov, 160, 120)
    a, b = small.render(160, 120, camera=cam), big.render(160, 120, camera=cam)
    assert np.array_equal(bits(a), bits(b))
    assert small.last_stats["rays_nearest"] == big.last_stats["rays_nearest"]
    small.close(); big.close()


@pytest.mark.parametrize("W,H,aa", [(1, 1, 1), (33, 9, 1), (500, 1, 1), (1, 70, 2), (40, 24, 4)])
def test_odd_frame_shapes(W, H, aa, gpu_scenes, flat_scenes):
    """Frames smaller than a tile, one pixel wide or high, with up to 16 samples per pixel; shards that own no tile at all."""
    import torch
    import raytracer_amd as rt
    fs = flat_scenes("c")
    sc = gpu_scenes("c")
    cam = rt.Camera(fs.fov, W, H)
    img = sc.render(W, H, aa=aa, camera=cam)
    ref, nn, ns = oracle.render(fs, cam, W, H, aa=aa, seed=1, threads=4)
    assert img.shape == (H, W, 3)
    assert np.abs(rgb8(img).astype(int) - rgb8(ref).astype(int)).max() <= 1
    assert sc.last_stats["rays_nearest"] == nn and sc.last_stats["rays_shadow"] == ns
    full = torch.zeros(H, W, 3, device="cuda")
    for rank in range(5):                                  # more ranks than tiles for the small frames
        sc.render_into(full.data_ptr(), W, H, aa=aa, camera=cam, tile_rank=rank, tile_ranks=5)
    assert np.array_equal(bits(full.cpu().numpy()), bits(img))


@pytest.mark.parametrize("name", ["a", "b", "e", "g", "h", "k", "random_5", "random_8"])
def test_pick_narration_vs_reference(name, gpu_scenes, flat_scenes):
    """trace(e, s, colour, pick=true): what the compiled reference prints on std::cout for 64 pixels of a 160x120 frame
    (tests/golden/pick_<scene>.txt, tools/make_pick_golden.py) -- hits, reflections, interior reflections of the exit search,
    lost rays, the depth limit -- against rt_trace_pick + rt_pick_narration."""
    import os
    import raytracer_amd as rt
    fs = flat_scenes(name)
    sc = gpu_scenes(name)
    W, H = 160, 120
    cam = rt.Camera(fs.fov, W, H)
    s = oracle.primary_targets(cam, W, H).reshape(H, W, 3)
    blocks = open(os.path.join(os.path.dirname(__file__), "golden", f"pick_{name}.txt")).read().split("== pixel ")[1:]
    assert len(blocks) == 64
    for b in blocks:
        head, _, rest = b.partition("\n")
        x, y = map(int, head.split())
        want, _, verdict = rest.rpartition("== ")
        hit, rgb, text = sc.trace_pick((0, 0, 0), s[y, x])
        assert hit == (verdict.strip() == "hit"), (x, y)
        assert text == want, f"pixel {x} {y}:\n{text}\n-- reference --\n{want}"


def test_reference_api_client_end_to_end(tmp_path):
    """A C++ caller written like q1.cpp (tests/shim_client.cpp: choose_scene, fov, background_colour, trace with and without
    pick, shadowRay) built against include/raytracer.h: its output equals what the C-ABI gives for the same rays."""
    import os
    import subprocess
    import raytracer_amd as rt
    from raytracer_amd import scenes
    from test_host_capi import build_shim_client
    exe = build_shim_client(tmp_path)
    os.makedirs(tmp_path / "scenes")
    path = scenes.random_scene(str(tmp_path / "scenes"), seed=19)
    name = os.path.basename(path)[:-5]
    W, H = 160, 120
    pixels = [(x, y) for y in range(10, H, 25) for x in range(12, W, 31)]
    out = subprocess.run([exe, name, str(W), str(H)] + [str(v) for p in pixels for v in p], cwd=str(tmp_path), check=True, capture_output=True, text=True).stdout
    flat = rt.FlatScene.from_json(path)
    sc = rt.Scene(flat, 0)
    cam = rt.Camera(flat.fov, W, H)
    s = oracle.primary_targets(cam, W, H).reshape(H, W, 3)
    # the three lines choose_scene prints (raytracer.cpp:122,138,142).  Their format is the compiled reference's own, observed on its
    # bundled scenes (oracle/_ref/ref_driver l: "Setting fov to 17 degrees." / "Setting background colour to vec3(0.000000, 0.000000, 0.100000)"):
    # operator<< of a double, and glm::to_string(vec3) = "vec3(%f, %f, %f)"
    bg = np.asarray(flat.background, np.float32)
    assert out.splitlines()[:3] == [f"Loading scene {name}", "Setting fov to 60 degrees.",
                                    "Setting background colour to vec3(%f, %f, %f)" % tuple(float(v) for v in bg)], out.splitlines()[:3]
    head = out[:out.index("== pixel")].splitlines()[-1].split()
    assert float(head[1]) == flat.fov and np.allclose([float(v) for v in head[3:6]], flat.background)
    blocks = out.split("== pixel ")[1:]
    assert len(blocks) == len(pixels)
    for (x, y), b in zip(pixels, blocks):
        lines = b.splitlines()
        assert lines[0].split() == [str(x), str(y)]
        k = next(i for i, l in enumerate(lines) if l.startswith("== "))
        narration = "".join(l + "\n" for l in lines[1:k])
        hit, rgb, text = sc.trace_pick((0, 0, 0), s[y, x])
        assert narration == text and (lines[k] == "== hit") == hit
        col = lines[k + 1].split()
        assert col[-1] == "1"                                     # pick and non-pick calls agree
        if hit:
            assert np.array_equal(np.array(col[1:4], np.float32), rgb)
            sh = lines[k + 2].split()
            r = sc.shadow_batch(s[y, x][None], np.array([[0.5, 6.0, -1.0]], np.float32))
            assert int(sh[1]) == int(r["lit"][0]) and np.array_equal(np.array(sh[2:5], np.float32), r["shadow"][0])
        else:
            assert col[1:4] == ["-1", "-1", "-1"]                 # colour untouched on a miss (raytracer.cpp:320-321)
    tail = lines[-2:] if lines[-1].startswith("beyond") else out.splitlines()[-2:]
    assert tail[0] == "Maximum number of reflections reached." and tail[1].split()[-4:] == ["0", "0", "0", "0"]
    sc.close()


def test_headless_renderer_cli(tmp_path, scene_dir, gpu_scenes, flat_scenes):
    """raytracer_amd/bin/rt_render (the headless stand-in for the GLUT front end): scene file in, BMP out -- the same
    pixels as the library's frame, quantised by q -> uint8(q * 255 + 0.5)."""
    import json
    import os
    import subprocess
    import raytracer_amd as rt
    from conftest import ROOT
    exe = os.path.join(ROOT, "raytracer_amd", "bin", "rt_render")
    if not os.path.exists(exe):
        pytest.skip("rt_render is not built (make cli)")
    out = str(tmp_path / "c.bmp")
    r = subprocess.run([exe, str(scene_dir / "c.rtscene"), "-o", out, "-w", "200", "-h", "100", "-aa", "2"], check=True, capture_output=True, text=True)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    fs, sc = flat_scenes("c"), gpu_scenes("c")
    img = sc.render(200, 100, aa=2, camera=rt.Camera(fs.fov, 200, 100))
    assert info["rays_nearest"] == sc.last_stats["rays_nearest"] and info["rays_shadow"] == sc.last_stats["rays_shadow"]
    data = open(out, "rb").read()
    assert data[:2] == b"BM" and int.from_bytes(data[18:22], "little") == 200 and int.from_bytes(data[22:26], "little") == 100
    off = int.from_bytes(data[10:14], "little")
    row = (200 * 3 + 3) // 4 * 4
    px = np.frombuffer(data, np.uint8, count=row * 100, offset=off).reshape(100, row)[:, :600].reshape(100, 200, 3)[:, :, ::-1]
    # BMP rows are stored bottom-up and the frame's row 0 is the bottom row (glDrawPixels order): same order
    assert np.array_equal(px, rgb8(img))
