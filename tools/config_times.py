#!/usr/bin/env python3
"""Frame times of the BASELINE.json configurations other than the bench one (C3, C4, C5 at a reduced triangle count)."""
import gzip, os, shutil, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes

def run(name, path, W, H, aa=(1, 1), max_depth=16, eye=(0, 0, 0), rot_x=0.0, reps=5):
    t0 = time.perf_counter()
    flat = rt.FlatScene.load(path) if path.endswith(".rtscene") else rt.FlatScene.from_json(path)
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    sc = rt.Scene(flat, 0)
    t_up = time.perf_counter() - t0
    cam = rt.Camera(flat.fov, W, H, eye=eye, rot_x=rot_x)
    frame = torch.zeros(H, W, 3, device="cuda")
    sc.render_into(frame.data_ptr(), W, H, aa=aa, camera=cam, max_depth=max_depth)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        st = sc.render_into(frame.data_ptr(), W, H, aa=aa, camera=cam, max_depth=max_depth)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    rays = st["rays_nearest"] + st["rays_shadow"]
    print(f"{name}: load {t_load:.1f} s, upload {t_up:.2f} s, {dt * 1e3:.2f} ms/frame, {rays / 1e6:.1f} Mrays ({st['rays_nearest'] / 1e6:.1f} nearest), {rays / dt / 1e6:.0f} Mrays/s, levels {st['levels']}, splits {st['batch_splits']}", flush=True)
    sc.close()

only = os.environ.get("CONFIG", "")
with tempfile.TemporaryDirectory() as tmp:
  if only in ("", "C1"):   # the reference's bundled scenes (tests/golden/scenes: the loader's flat form of scenes/a..m.json) at its own 640x480
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
    for f in sorted(os.listdir(gold)):
      if f.endswith(".rtscene.gz") and f[0] in os.environ.get("C1_SCENES", "abcdefghijklm"):
        dst = os.path.join(tmp, f[:-3])
        with gzip.open(os.path.join(gold, f), "rb") as a, open(dst, "wb") as b:
          shutil.copyfileobj(a, b)
        run(f"C1 scene {f[0]} 640x480", dst, 640, 480, reps=10)
  if only in ("", "C3"):
    run("C3 csg 1080p depth5", scenes.c3_scene(tmp), 1920, 1080, max_depth=5, eye=(0, 1.0, 2.0), rot_x=-0.25)
  if only in ("", "C4"):
    run("C4 area 64 1080p", scenes.c4_scene(tmp, 64), 1920, 1080)
  g = int(os.environ.get("C5_GRID", "500"))
  if only in ("", "C5"):
    a = int(os.environ.get("C5_AA", "2"))   # BASELINE.json quotes C5 at 4x4 samples per pixel on 8 GPUs; 2x2 is one GPU's quarter of that
    run(f"C5 grid {g} 4K {a}x{a}", scenes.c5_scene(tmp, g), 3840, 2160, aa=(a, a), eye=(0, 1.5, 1.0), rot_x=-0.3, reps=2)
