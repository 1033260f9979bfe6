import gzip, os, shutil, sys, tempfile
sys.path.insert(0, '/root/repo')
import raytracer_amd as rt
from raytracer_amd import scenes
with tempfile.TemporaryDirectory() as tmp:
    for s in "abcdefghijklm":
        dst = os.path.join(tmp, f"{s}.rtscene")
        with gzip.open(f"/root/repo/tests/golden/scenes/{s}.rtscene.gz", "rb") as a, open(dst, "wb") as b: shutil.copyfileobj(a, b)
        f = rt.FlatScene.load(dst); i = f.info
        print(s, {k: getattr(i, k) for k in dir(i) if k.startswith('n_')}, flush=True)
        sc = rt.Scene(f, 0); sc.close()
    f = rt.FlatScene.from_json(scenes.c3_scene(tmp)); i = f.info
    print('C3', {k: getattr(i, k) for k in dir(i) if k.startswith('n_')}, flush=True)
    sc = rt.Scene(f, 0); sc.close()
