import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes
def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
seed, n = int(sys.argv[1]), int(sys.argv[2])
W, H, depth = 256, 144, 6
with tempfile.TemporaryDirectory() as tmp:
    path = scenes.random_scene(tmp, seed=seed, n_objects=n)
    doc = json.load(open(path))
    for o in doc["objects"]:
        if o["type"] == "texturemesh": o["material"].pop("reflective", None)
    json.dump(doc, open(path, "w"))
    flat = rt.FlatScene.from_json(path)
def kinds(o):
    if o["type"] != "csgobject": return o["type"]
    def rec(t): return t["type"] if "type" in t and "operation" not in t else f"{t['operation'][0]}({rec(t['first'])},{rec(t['second'])})"
    return "csg:" + rec({k: v for k, v in o.items() if k in ("operation", "first", "second")})
print([ (kinds(o), "R" if "reflective" in o["material"] else "", "T%s" % o["material"].get("refraction") if "transmissive" in o["material"] else "") for o in doc["objects"]])
print([l["type"] for l in doc["lights"]], {k: getattr(flat.info, k) for k in dir(flat.info) if k.startswith("n_")})
os.environ["RT_LOAD_TIMING"] = "1"
sc = rt.Scene(flat, 0)
cam = rt.Camera(flat.fov, W, H, eye=(0, 1.0, 1.5), rot_x=-0.2)
sw = ["RT_NO_FUSED_LEVELS", "RT_NO_SMALL_SCENE", "RT_NO_SMALL_LEVEL0", "RT_NO_INLINE_TRANSMIT", "RT_NO_SELF_REPORT"]
for v in sw: os.environ.pop(v, None)
os.environ["RT_NO_FUSED_LEVELS"] = "1"
old = sc.render(W, H, camera=cam, max_depth=depth, seed=7); so = dict(sc.last_stats)
os.environ.pop("RT_NO_FUSED_LEVELS")
for v in [None] + sw[1:]:
    if v: os.environ[v] = "1"
    img = sc.render(W, H, camera=cam, max_depth=depth, seed=7); st = dict(sc.last_stats)
    if v: os.environ.pop(v)
    d = np.abs(img - old).max(axis=2)
    ys, xs = np.nonzero(d > 0)
    print(v, "equal" if np.array_equal(bits(img), bits(old)) else f"{len(ys)} pixels differ max {d.max():.3g} first {list(zip(xs[:3], ys[:3]))}", (st["rays_nearest"], st["rays_shadow"], st["levels"]), (so["rays_nearest"], so["rays_shadow"], so["levels"]))
sc.close()
