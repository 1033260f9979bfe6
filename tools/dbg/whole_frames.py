"""Hypothesis test: whole frames per lane instead of a shard per lane.  NS scene handles, each with ONE lane (RT_LANES=1) and its own stream,
frames dealt round-robin, D in flight per handle; against the default (one handle, 4 lanes, every frame cut into 4 shards).
usage: python tools/dbg/whole_frames.py [NS] [D] [frames]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
D = int(sys.argv[2]) if len(sys.argv) > 2 else 2
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 400
os.environ["RT_LANES"] = "1"
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H = 1920, 1080
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.c2_scene(tmp, grid=224))
cam = rt.Camera(flat.fov, W, H, eye=(0.0, 1.5, 1.0), rot_x=-0.3)
scs = [rt.Scene(flat, 0) for _ in range(NS)]
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(NS - 1)]   # (handle 0 on the caller's stream: the process has 4 hardware queues)
bufs = [[torch.zeros(H, W, 3, device="cuda") for _ in range(D)] for _ in range(NS)]
pend = [[] for _ in range(NS)]
def run(n):
    rays = 0
    for i in range(n):
        k = i % NS
        if len(pend[k]) == D:
            st = scs[k].render_wait(pend[k].pop(0)); rays += st["rays_nearest"] + st["rays_shadow"]
        pend[k].append(scs[k].render_submit(bufs[k][(i // NS) % D].data_ptr(), W, H, camera=cam, stream=streams[k].cuda_stream))
    for k in range(NS):
        while pend[k]:
            st = scs[k].render_wait(pend[k].pop(0)); rays += st["rays_nearest"] + st["rays_shadow"]
    return rays
run(20); torch.cuda.synchronize()
t0 = time.perf_counter(); rays = run(frames); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{NS} handles x 1 lane, {D} in flight each: {dt / frames * 1e3:.4f} ms/frame, {rays / dt / 1e6:.0f} Mrays/s", flush=True)
for s in scs: s.close()
