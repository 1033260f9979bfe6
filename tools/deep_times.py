#!/usr/bin/env python3
"""GPU frame times (3 frames in flight, and one alone) of the configurations whose ray trees go deep: C3 and the bundled scenes
a b e g h i l at 640x480 -- the per-level floor VERDICT r02 item 2 is about.  SCENES=C3,h,i restricts; INFLIGHT=n frames in flight."""
import gzip, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracer_amd as rt
from raytracer_amd import scenes

D = int(os.environ.get("INFLIGHT", "3"))
FLAGS = rt_flags = int(os.environ.get("WHOLE", "0")) * 4   # WHOLE=1: RT_FLAG_WHOLE_FRAME (use INFLIGHT=6 with it)
AA = int(os.environ.get("AA", "1"))   # AA x AA samples per pixel


def run(name, flat, W, H, max_depth=16, eye=(0, 0, 0), rot_x=0.0, reps=20):
    sc = rt.Scene(flat, 0)
    cam = rt.Camera(flat.fov, W, H, eye=eye, rot_x=rot_x)
    bufs = [torch.zeros(H, W, 3, device="cuda") for _ in range(D)]
    for _ in range(2):
        st = sc.render_into(bufs[0].data_ptr(), W, H, camera=cam, max_depth=max_depth, aa=(AA, AA))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        st = sc.render_into(bufs[0].data_ptr(), W, H, camera=cam, max_depth=max_depth, aa=(AA, AA))
    torch.cuda.synchronize(); alone = (time.perf_counter() - t0) / reps
    pend, n = [], D * reps
    t0 = time.perf_counter()
    for i in range(n):
        if len(pend) == D:
            sc.render_wait(pend.pop(0))
        pend.append(sc.render_submit(bufs[i % D].data_ptr(), W, H, camera=cam, max_depth=max_depth, aa=(AA, AA), flags=FLAGS))
    for t in pend:
        sc.render_wait(t)
    torch.cuda.synchronize(); piped = (time.perf_counter() - t0) / n
    rays = st["rays_nearest"] + st["rays_shadow"]
    print(f"{name}: {piped * 1e3:.4f} ms/frame piped ({rays / piped / 1e6:.0f} Mrays/s), {alone * 1e3:.4f} alone, levels {st['levels']}", flush=True)
    sc.close()


only = [s for s in os.environ.get("SCENES", "C3,a,b,e,g,h,i,l").split(",") if s]
with tempfile.TemporaryDirectory() as tmp:
    for s in only:
        if s == "C3":
            run("C3", rt.FlatScene.from_json(scenes.c3_scene(tmp)), 1920, 1080, max_depth=5, eye=(0, 1.0, 2.0), rot_x=-0.25, reps=10)
        elif s == "C2":
            run("C2", rt.FlatScene.from_json(scenes.c2_scene(tmp, grid=224)), 1920, 1080, eye=(0.0, 1.5, 1.0), rot_x=-0.3, reps=30)
        else:
            dst = os.path.join(tmp, f"{s}.rtscene")
            with gzip.open(os.path.join(ROOT, "tests", "golden", "scenes", f"{s}.rtscene.gz"), "rb") as a, open(dst, "wb") as b:
                shutil.copyfileobj(a, b)
            run(f"scene {s}", rt.FlatScene.load(dst), 640, 480)
