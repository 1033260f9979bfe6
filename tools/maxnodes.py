import tempfile, sys
sys.path.insert(0,'.')
import raytracer_amd as rt
from raytracer_amd import scenes
import torch
with tempfile.TemporaryDirectory() as tmp:
    flat = rt.FlatScene.from_json(scenes.c2_scene(tmp))
sc = rt.Scene(flat, 0)
cam = rt.Camera(flat.fov, 1920, 1080, eye=(0,1.5,1.0), rot_x=-0.3)
frame = torch.zeros(1080,1920,3,device='cuda')
st = sc.render_into(frame.data_ptr(), 1920, 1080, camera=cam, flags=rt.RT_FLAG_COUNTERS)
from raytracer_amd.capi import rt_stats
print(st)
