import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RT_LOAD_TIMING"] = "1"
import torch, raytracer_amd as rt
from raytracer_amd import scenes
with tempfile.TemporaryDirectory() as tmp:
    for name, path in (("C2", scenes.c2_scene(tmp, grid=224)), ("C5", scenes.c5_scene(tmp, 1581)), ("C4", scenes.c4_scene(tmp, 64)), ("C3", scenes.c3_scene(tmp))):
        print(name, flush=True)
        flat = rt.FlatScene.from_json(path); sc = rt.Scene(flat, 0); sc.close()
