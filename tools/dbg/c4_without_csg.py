#!/usr/bin/env python3
"""C4 with its two CSG objects replaced by their first operands (a box and a sphere): what the area-light kernel costs when the scene
holds no CSG object, i.e. in its variant without csgEval (the rays are nearly the same)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
with tempfile.TemporaryDirectory() as tmp:
    path = scenes.c4_scene(tmp, 64)
    doc = json.load(open(path))
    base = json.loads(json.dumps(doc))
    def shifted(node, dy):   # a CSG tree moved by dy
        n = dict(node)
        if "operation" in n: n["first"] = shifted(n["first"], dy); n["second"] = shifted(n["second"], dy)
        elif n["type"] == "sphere": n["position"] = [n["position"][0], n["position"][1] + dy, n["position"][2]]
        else: n["point1"] = [n["point1"][0], n["point1"][1] + dy, n["point1"][2]]; n["point2"] = [n["point2"][0], n["point2"][1] + dy, n["point2"][2]]
        return n
    for variant in ("as is", "CSG objects replaced by their first operands", "the same, plus the CSG objects far below the floor (the kernels' CSG variants, no evaluation)"):
        if variant != "as is":
            objs = []
            for o in base["objects"]:
                if o["type"] == "csgobject":
                    f = dict(o["first"]); f["material"] = o["material"]; objs.append(f)
                    if variant.startswith("the same"): objs.append(shifted(o, -500.0))
                else: objs.append(o)
            doc = dict(base); doc["objects"] = objs
            path = os.path.join(tmp, f"c4_{len(variant)}.json"); json.dump(doc, open(path, "w"))
        flat = rt.FlatScene.from_json(path)
        sc = rt.Scene(flat, 0)
        W, H = 1920, 1080
        frame = torch.zeros(H, W, 3, device="cuda")
        sc.render_into(frame.data_ptr(), W, H)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): st = sc.render_into(frame.data_ptr(), W, H)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        rays = st["rays_nearest"] + st["rays_shadow"]
        print(f"C4 {variant}: {dt * 1e3:.2f} ms/frame, {rays / 1e6:.1f} Mrays, {rays / dt / 1e6:.0f} Mrays/s", flush=True)
        sc.close()
