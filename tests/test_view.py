"""The reference's interactive front end without its window (rt_view_*, q1.cpp:107-288): the key handling, the pose and the scanline
order need no GPU and are checked here against q1.cpp's own arithmetic written out; rendering and the mouse pick are in the GPU part."""
import math

import numpy as np
import pytest

import raytracer_amd as rt

f32 = np.float32
MOVE, ROT = f32(0.5), f32(math.pi / 8)          # q1.cpp:27-28


def test_keys_move_the_camera_as_keyboard_does():
    v = rt.View(None, fov=60.0, width=640, height=480)
    st = v.state()
    assert st["needs_redraw"] and np.array_equal(st["eye"], [0, 0, 0]) and st["rotation_x"] == 0 and st["rotation_y"] == 0
    cam, aa = v.camera()
    assert aa == 1 and np.allclose(cam.c.facing[:], [0, 0, -1])
    # w / s along facing, a / d along the normalised right vector, t / g along y: all by move_step (q1.cpp:189-226)
    assert v.key("w") == (1, "")
    assert np.array_equal(v.state()["eye"], np.array([0, 0, -0.5], f32))
    def glm_normalize(a):                       # v * (1 / sqrt(dot(v, v))) in float, glm/detail/func_geometric.inl:88-96
        a = np.asarray(a, f32)
        t = a * a
        return a * (f32(1) / np.sqrt(f32(f32(t[0] + t[1]) + t[2])))
    step = glm_normalize(cam.c.right[:]) * MOVE  # eye += normalize(camera_right) * move_step (camera_right is scaled by the window's w)
    v.key("d")
    assert np.array_equal(v.state()["eye"].view(np.uint32), (np.array([0, 0, -0.5], f32) + step).view(np.uint32))
    v.keys("sat")
    assert np.array_equal(v.state()["eye"], np.array([0, 0.5, 0], f32))
    v.key("g")
    # q / e turn by pi/8 in float (q1.cpp:205-212); the pose of the NEXT frame follows, what w moves along is the frame on screen
    v.key("q")
    assert v.state()["rotation_y"] == float(ROT)
    v.key("w")
    assert np.array_equal(v.state()["eye"], np.array([0, 0, -0.5], f32))          # still along the old facing: no display() in between
    ref = rt.Camera(60.0, 640, 480, eye=(0, 0, -0.5), rot_y=float(ROT))
    cam, _ = v.camera()
    for name in ("eye", "facing", "right", "up"):
        assert np.array_equal(np.array(getattr(cam.c, name)[:], f32).view(np.uint32), np.array(getattr(ref.c, name)[:], f32).view(np.uint32))
    # r is limited to rotationX < pi/2 BEFORE the step (q1.cpp:213-218): four steps reach pi/2 in float, which is still below the double
    # pi/2 or not as float rounding has it -- the reference's own comparison decides
    rx, steps = f32(0), 0
    while float(rx) < math.pi / 2:
        rx = f32(rx + ROT); steps += 1
    for _ in range(8):
        v.key("r")
    assert v.state()["rotation_x"] == float(rx) and steps in (4, 5)
    assert v.key("r")[0] == 0                                                      # at the limit: no new frame
    for _ in range(20):
        v.key("f")
    lo = f32(rx)
    while float(lo) > -math.pi / 2:
        lo = f32(lo - ROT)
    assert v.state()["rotation_x"] == float(lo)
    # l toggles the four-sample supersampling and says so; p prints camera and facing; space and unknown keys change nothing; Escape = -1
    assert v.key("l") == (1, "Anti-aliasing ON\n") and v.camera()[1] == 2
    assert v.key("l") == (1, "Anti-aliasing OFF\n") and v.camera()[1] == 1
    rc, text = v.key("p")
    assert rc == 0 and text.startswith("camera: (0, 0, -0.5)\nfacing: <0, 0, -1>\n")
    assert v.key(" ")[0] == 0 and v.key("x")[0] == 0 and v.key(27)[0] == -1
    v.reshape(320, 200)
    assert (v.state()["width"], v.state()["height"]) == (320, 200) and v.state()["needs_redraw"]
    v.close()


@pytest.mark.parametrize("height", [480, 512, 100, 17, 1080])
def test_scanline_order_is_displays(height):
    """q1.cpp:124: y = ((int(drawing_y) - 1) * 16 % vp_height) + (((int(drawing_y) - 1) * 16) / vp_height) * 7 % 16 for drawing_y = 1 .. vp_height."""
    rows = [rt.View.scanline(height, k) for k in range(height)]
    want = [((k * 16) % height) + (((k * 16) // height) * 7) % 16 for k in range(height)]
    assert rows == want
    assert rt.View.scanline(height, -1) == -1 and rt.View.scanline(0, 0) == -1
    if height % 16 == 0:
        assert sorted(rows) == list(range(height))          # the 16 sweeps fill a window whose height is a multiple of 16 exactly once


@pytest.mark.gpu
def test_view_renders_what_rt_render_renders_and_click_narrates(gpu_scenes, flat_scenes, tmp_path):
    import os
    import subprocess
    import torch
    from conftest import GOLD, ROOT
    sc, fs = gpu_scenes("g"), flat_scenes("g")
    W, H = 320, 240
    v = rt.View(sc, width=W, height=H)
    buf = torch.zeros(H, W, 3, device="cuda")
    st = v.render_into(buf.data_ptr())
    want = sc.render(W, H, camera=rt.Camera(fs.fov, W, H))
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), want.view(np.uint32)) and st["rays_nearest"] >= W * H and not v.state()["needs_redraw"]
    # moved, turned, supersampled: display() after "wdqrl" is rt_render with that pose and the 2x2 grid
    v.keys("wdqrl")
    frame = v.render()
    stt = v.state()
    cam = rt.Camera(fs.fov, W, H, eye=tuple(stt["eye"]), rot_x=stt["rotation_x"], rot_y=stt["rotation_y"])
    assert np.array_equal(frame.view(np.uint32), sc.render(W, H, aa=2, camera=cam).view(np.uint32))
    # a click: the narration of trace(..., pick = true) for that pixel, then the HIT / MISS lines of q1.cpp:258-266
    v.keys("l")
    v.render()
    x, y_top = 150, 130
    text = v.click(x, y_top)
    s = (np.array(cam.c.eye[:], f32) + np.array(cam.c.facing[:], f32)
         + np.array(cam.c.right[:], f32) * f32(2 * ((x + f32(0.5)) / f32(W) - f32(0.5)))
         + np.array(cam.c.up[:], f32) * f32(2 * ((H - y_top - 1 + f32(0.5)) / f32(H) - f32(0.5))))
    hit, rgb, narration = sc.trace_pick(np.array(cam.c.eye[:], f32), s)
    assert text.startswith("\n" + narration) and (("HIT @ ( " in text and "      colour = ( " in text) if hit else "MISS @ ( " in text) and text.endswith("\n\n")
    # a move key and then a click WITHOUT a frame in between: mouse() -> s(x, y) and trace(eye, ...) read the global eye, which `w` has
    # already moved, together with the facing / camera_right / camera_up of the frame still on screen (q1.cpp:51-53, 189-191, 251-259)
    v.key("w")
    eye2 = v.state()["eye"].astype(f32)
    assert not np.array_equal(eye2, np.array(cam.c.eye[:], f32))
    text2 = v.click(x, y_top)
    s2 = (eye2 + np.array(cam.c.facing[:], f32)
          + np.array(cam.c.right[:], f32) * f32(2 * ((x + f32(0.5)) / f32(W) - f32(0.5)))
          + np.array(cam.c.up[:], f32) * f32(2 * ((H - y_top - 1 + f32(0.5)) / f32(H) - f32(0.5))))
    hit2, rgb2, narration2 = sc.trace_pick(eye2, s2)
    want_at = ("HIT @ ( " if hit2 else "MISS @ ( ") + ",".join("%g" % c for c in s2) + " )\n"
    assert text2.startswith("\n" + narration2) and want_at in text2, (text2, want_at)
    v.close()
    # the scripted session: frames as BMPs, the sixteen sweeps of the first one, the print-outs of p and l
    scenes_dir = tmp_path / "scenes"
    scenes_dir.mkdir()
    import gzip, shutil
    with gzip.open(os.path.join(GOLD, "scenes", "c.rtscene.gz"), "rb") as a, open(tmp_path / "c.rtscene", "wb") as b:
        shutil.copyfileobj(a, b)
    exe = os.path.join(ROOT, "raytracer_amd", "bin", "rt_view")
    r = subprocess.run([exe, str(tmp_path / "c.rtscene"), "-w", "160", "-h", "128", "-o", str(tmp_path / "f"), "-progressive", ".wwq.lp.c80,64;"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("[frame ") == 3 and "Anti-aliasing ON" in r.stdout and "camera: (0, 0, -1)" in r.stdout and ("HIT @" in r.stdout or "MISS @" in r.stdout)
    for name in ("f000.bmp", "f001.bmp", "f002.bmp", "f000_sweep00.bmp", "f000_sweep15.bmp"):
        assert os.path.getsize(tmp_path / name) == 54 + 160 * 128 * 3
    assert open(tmp_path / "f000_sweep15.bmp", "rb").read() == open(tmp_path / "f000.bmp", "rb").read()      # after the last sweep the window is full
