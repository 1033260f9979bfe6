"""Deterministic synthetic scenes for the BASELINE.json configs, written in the reference's own
JSON scene schema (raytracer.cpp:134-300) plus procedurally generated 24-bit BMP textures, so they go through the
same loader as the bundled scenes.  Nothing here is read from the reference at run time.

    C2  ~100k-triangle textured height field + two bump-mapped spheres   (the 1080p headline workload)
    C3  8x8 lattice of nested CSG objects, reflective / refractive, depth 5
    C4  area-light soft shadows, 64 samples per light
    C5  ~5M-triangle height field (generated at reduced size unless asked; see c5_scene)
"""
import json
import math
import os
import struct

import numpy as np


def write_bmp24(path, rgb):
    """rgb: uint8 [H, W, 3], row 0 = top.  Uncompressed 24-bit BMP (what EasyBMP reads, EasyBMP.cpp:668)."""
    h, w = rgb.shape[:2]
    row = (w * 3 + 3) // 4 * 4
    data = np.zeros((h, row), np.uint8)
    data[:, : w * 3] = rgb[::-1, :, ::-1].reshape(h, w * 3)   # bottom-up rows, BGR
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + row * h, 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, row * h, 3780, 3780, 0, 0))
        f.write(data.tobytes())


def _hash01(ix, iz, seed):
    """Integer lattice hash -> [0,1), vectorised (the 'hash-noise' of the height fields)."""
    h = (ix.astype(np.uint64) * np.uint64(0x9E3779B1) + iz.astype(np.uint64) * np.uint64(0x85EBCA77) + np.uint64(seed) * np.uint64(0xC2B2AE3D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return (h & np.uint64(0xFFFFFF)).astype(np.float64) / float(1 << 24)


def tiles_texture(size=1024, cells=16):
    y, x = np.mgrid[0:size, 0:size]
    c = ((x * cells // size) + (y * cells // size)) % 2
    grout = ((x % (size // cells)) < 3) | ((y % (size // cells)) < 3)
    img = np.zeros((size, size, 3), np.uint8)
    img[..., 0] = np.where(c, 200, 60) + (x * 40 // size)
    img[..., 1] = np.where(c, 180, 70) + (y * 50 // size)
    img[..., 2] = np.where(c, 150, 110)
    img[grout] = (30, 30, 35)
    return img


def bump_texture(size=512, seed=7):
    y, x = np.mgrid[0:size, 0:size].astype(np.float64)
    v = np.zeros((size, size))
    for octave, amp in ((8, 0.5), (16, 0.3), (32, 0.2)):
        gx, gy = x * octave / size, y * octave / size
        ix, iy = np.floor(gx).astype(np.int64), np.floor(gy).astype(np.int64)
        fx, fy = gx - ix, gy - iy
        fx, fy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
        a = _hash01(ix % octave, iy % octave, seed + octave)
        b = _hash01((ix + 1) % octave, iy % octave, seed + octave)
        c = _hash01(ix % octave, (iy + 1) % octave, seed + octave)
        d = _hash01((ix + 1) % octave, (iy + 1) % octave, seed + octave)
        v += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
    g = np.clip(v * 255, 0, 255).astype(np.uint8)
    return np.stack([g, g, g], axis=-1)


def _fmt(v):
    return float(np.float32(v))


def height_field(n, seed, x0=-8.0, x1=8.0, z0=-17.0, z1=-1.0):
    """n x n cells -> (triangles [2n^2, 3, 3], uv [2n^2, 3, 2]); normals face +y (triangles are back-face culled)."""
    ix, iz = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    x = x0 + (x1 - x0) * ix / n
    z = z0 + (z1 - z0) * iz / n
    y = -1.2 + 0.3 * np.sin(1.7 * x) * np.cos(1.3 * z) + _hash01(ix, iz, seed) * 0.02
    P = np.stack([x, y, z], axis=-1).astype(np.float32)            # [n+1, n+1, 3]
    UV = np.stack([ix / n, iz / n], axis=-1).astype(np.float32)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
    ua, ub, uc, ud = UV[:-1, :-1], UV[1:, :-1], UV[:-1, 1:], UV[1:, 1:]
    tris = np.stack([np.stack([a, c, b], axis=2), np.stack([b, c, d], axis=2)], axis=2).reshape(-1, 3, 3)
    uvs = np.stack([np.stack([ua, uc, ub], axis=2), np.stack([ub, uc, ud], axis=2)], axis=2).reshape(-1, 3, 2)
    return tris, uvs


def _mesh_json(tris, uvs=None):
    out = {"triangles": tris.astype(np.float64).round(7).tolist()}
    if uvs is not None:
        out["uvCoords"] = uvs.astype(np.float64).round(7).tolist()
    return out


def _mesh_file(out_dir, name, tris, uvs=None):
    """The same triangles as _mesh_json, as the loader's binary mesh file (scene_loader.cpp, MeshData): {"file": name}.
    Values go through the same 7-decimal rounding as the JSON text, so both forms load to bit-identical scenes."""
    pos = tris.astype(np.float64).round(7).astype("<f4")
    with open(os.path.join(out_dir, name), "wb") as f:
        f.write(b"RTMESH01")
        f.write(np.array([len(pos)], "<u8").tobytes())
        f.write(np.array([1 if uvs is not None else 0, 0], "<u4").tobytes())
        f.write(np.ascontiguousarray(pos).tobytes())
        if uvs is not None:
            f.write(np.ascontiguousarray(uvs.astype(np.float64).round(7).astype("<f4")).tobytes())
    return {"file": name}


def c2_scene(out_dir, grid=224):
    """BASELINE config 2: textured triangle mesh (grid*grid*2 triangles, 224 -> 100 352) with bump mapping."""
    os.makedirs(out_dir, exist_ok=True)
    write_bmp24(os.path.join(out_dir, "gen_tiles.bmp"), tiles_texture())
    write_bmp24(os.path.join(out_dir, "gen_bump1.bmp"), bump_texture(seed=7))
    write_bmp24(os.path.join(out_dir, "gen_bump2.bmp"), bump_texture(seed=21))
    tris, uvs = height_field(grid, seed=1)
    mesh = {"type": "texturemesh", "texture": "gen_tiles.bmp",
            "material": {"specular": [0.2, 0.2, 0.2], "shininess": 12.0}}
    mesh.update(_mesh_json(tris, uvs))
    scene = {
        "camera": {"field": 60, "background": [0.25, 0.35, 0.55]},
        "objects": [
            mesh,
            {"type": "bumpsphere", "radius": 1.0, "position": [-1.6, -0.1, -6.5], "bumpmap": "gen_bump1.bmp", "bumpdepth": 2.0,
             "material": {"ambient": [0.2, 0.1, 0.1], "diffuse": [0.8, 0.3, 0.2], "specular": [0.5, 0.5, 0.5], "shininess": 20.0}},
            {"type": "bumpsphere", "radius": 1.2, "position": [1.7, 0.2, -8.0], "bumpmap": "gen_bump2.bmp", "bumpdepth": 3.0,
             "material": {"ambient": [0.1, 0.1, 0.2], "diffuse": [0.2, 0.4, 0.8], "specular": [0.6, 0.6, 0.6], "shininess": 40.0}},
        ],
        "lights": [
            {"type": "ambient", "color": [0.25, 0.25, 0.25]},
            {"type": "directional", "color": [0.7, 0.7, 0.65], "direction": [-0.4, -1.0, -0.3]},
            {"type": "point", "color": [0.5, 0.45, 0.4], "position": [2.0, 4.0, -3.0]},
        ],
    }
    path = os.path.join(out_dir, f"c2_{grid}.json")
    with open(path, "w") as f:
        json.dump(scene, f)
    return path


def _csg_cell(cx, cz, s):
    """The k.json pattern: difference(intersection(box, sphere), union(box, union(box, box))) -- 5 leaves, depth 4."""
    def box(ax, ay, az, bx, by, bz):
        return {"type": "box", "point1": [_fmt(cx + ax * s), _fmt(-1 + ay * s), _fmt(cz + az * s)], "point2": [_fmt(cx + bx * s), _fmt(-1 + by * s), _fmt(cz + bz * s)]}
    return {
        "operation": "difference",
        "first": {"operation": "intersection", "first": box(-1, 0, -1, 1, 2, 1),
                  "second": {"type": "sphere", "position": [_fmt(cx), _fmt(-1 + s), _fmt(cz)], "radius": _fmt(1.35 * s)}},
        "second": {"operation": "union", "first": box(-0.5, 0.5, -1.5, 0.5, 1.5, 1.5),
                   "second": {"operation": "union", "first": box(-1.5, 0.5, -0.5, 1.5, 1.5, 0.5), "second": box(-0.5, -0.5, -0.5, 0.5, 2.5, 0.5)}},
    }


def c3_scene(out_dir):
    """BASELINE config 3: 64 nested CSG objects on an 8x8 lattice; half reflective 0.5, a quarter transmissive 0.6 / ior 1.33."""
    os.makedirs(out_dir, exist_ok=True)
    objects = [{"type": "plane", "position": [0, -1, 0], "normal": [0, 1, 0],
                "material": {"ambient": [0.1, 0.1, 0.1], "diffuse": [0.5, 0.5, 0.45], "reflective": [0.2, 0.2, 0.2]}}]
    for i in range(8):
        for j in range(8):
            cx, cz = -5.25 + 1.5 * i, -4.0 - 1.5 * j
            k = i * 8 + j
            mat = {"ambient": [0.1 + 0.05 * (k % 3), 0.1, 0.15], "diffuse": [0.3 + 0.08 * (i % 5), 0.3 + 0.08 * (j % 5), 0.4],
                   "specular": [0.5, 0.5, 0.5], "shininess": 30.0}
            if k % 2 == 0:
                mat["reflective"] = [0.5, 0.5, 0.5]
            if k % 4 == 1:
                mat["transmissive"] = [0.6, 0.6, 0.6]
                mat["refraction"] = 1.33
            obj = {"type": "csgobject", "material": mat}
            obj.update(_csg_cell(cx, cz, 0.45))
            objects.append(obj)
    scene = {"camera": {"field": 60, "background": [0.1, 0.12, 0.2]}, "objects": objects,
             "lights": [{"type": "ambient", "color": [0.3, 0.3, 0.3]},
                        {"type": "directional", "color": [0.8, 0.8, 0.75], "direction": [-0.3, -1.0, -0.5]}]}
    path = os.path.join(out_dir, "c3.json")
    with open(path, "w") as f:
        json.dump(scene, f)
    return path


def random_csg_scene(out_dir, seed=0, n_objects=24, max_leaves=16):
    """Random CSG programs for the parity tests: `n_objects` csgobjects on a lattice, each a random binary tree of
    union / intersection / difference over 2..max_leaves overlapping spheres and boxes (multi-interval lists, empty
    operands, deep stacks), a third of them reflective and a third transmissive."""
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)

    def leaf(cx, cy, cz, r):
        ox, oy, oz = (float(v) for v in rng.uniform(-0.6, 0.6, 3) * r)
        if rng.random() < 0.5:
            return {"type": "sphere", "position": [_fmt(cx + ox), _fmt(cy + oy), _fmt(cz + oz)], "radius": _fmt(float(rng.uniform(0.35, 0.8)) * r)}
        hx, hy, hz = (float(v) for v in rng.uniform(0.25, 0.8, 3) * r)
        return {"type": "box", "point1": [_fmt(cx + ox - hx), _fmt(cy + oy - hy), _fmt(cz + oz - hz)],
                "point2": [_fmt(cx + ox + hx), _fmt(cy + oy + hy), _fmt(cz + oz + hz)]}

    def tree(leaves, cx, cy, cz, r):
        if leaves == 1:
            return leaf(cx, cy, cz, r)
        left = int(rng.integers(1, leaves))
        return {"operation": str(rng.choice(["union", "union", "intersection", "difference"])),
                "first": tree(left, cx, cy, cz, r), "second": tree(leaves - left, cx, cy, cz, r)}

    objects = [{"type": "plane", "position": [0, -1.2, 0], "normal": [0, 1, 0],
                "material": {"ambient": [0.1, 0.1, 0.1], "diffuse": [0.5, 0.5, 0.45], "reflective": [0.15, 0.15, 0.15]}}]
    cols = int(math.ceil(math.sqrt(n_objects)))
    for k in range(n_objects):
        i, j = k % cols, k // cols
        cx, cy, cz = -1.6 * (cols - 1) / 2 + 1.6 * i, float(rng.uniform(-0.5, 0.6)), -4.0 - 1.6 * j
        mat = {"ambient": [0.1, 0.1, 0.12], "diffuse": [_fmt(float(v)) for v in rng.uniform(0.2, 0.8, 3)],
               "specular": [0.4, 0.4, 0.4], "shininess": 20.0}
        if k % 3 == 0:
            mat["reflective"] = [0.4, 0.4, 0.4]
        if k % 3 == 1:
            mat["transmissive"] = [0.5, 0.5, 0.5]
            mat["refraction"] = [1.0, 1.33, 1.5, 0.0][(k // 3) % 4]
        obj = {"type": "csgobject", "material": mat}
        obj.update(tree(int(rng.integers(2, max_leaves + 1)), cx, cy, cz, 0.7))
        objects.append(obj)
    scene = {"camera": {"field": 60, "background": [0.1, 0.12, 0.2]}, "objects": objects,
             "lights": [{"type": "ambient", "color": [0.3, 0.3, 0.3]},
                        {"type": "point", "color": [0.7, 0.7, 0.7], "position": [1.0, 4.0, 0.0]},
                        {"type": "directional", "color": [0.4, 0.4, 0.4], "direction": [-0.3, -1.0, -0.5]}]}
    path = os.path.join(out_dir, f"csg_random_{seed}.json")
    with open(path, "w") as f:
        json.dump(scene, f)
    return path


def random_scene(out_dir, seed=0, n_objects=40):
    """A random mix of everything the trace path handles, for fuzzing parity against the oracle: planes, spheres, boxes,
    small and mid-size meshes (one of them textured, one a CSG operand), bump-mapped spheres, CSG trees; materials that
    reflect, transmit (ior 0 / 1 / 1.33 / 1.5) or both; ambient, directional, point and spot lights and sometimes a
    small area light."""
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(1000 + seed)
    write_bmp24(os.path.join(out_dir, "fz_tex.bmp"), tiles_texture(64, 4))
    write_bmp24(os.path.join(out_dir, "fz_bump.bmp"), bump_texture(64, seed=seed + 1))

    def vec(lo, hi):
        return [_fmt(float(v)) for v in rng.uniform(lo, hi, 3)]

    def material():
        m = {"ambient": vec(0.0, 0.2), "diffuse": vec(0.1, 0.9)}
        if rng.random() < 0.6:
            m["specular"] = vec(0.1, 0.8); m["shininess"] = _fmt(float(rng.uniform(2, 60)))
        r = rng.random()
        if r < 0.25 or r > 0.9:
            m["reflective"] = vec(0.1, 0.7)
        if r > 0.7:
            m["transmissive"] = vec(0.2, 0.8)
            m["refraction"] = [0.0, 1.0, 1.33, 1.5][int(rng.integers(0, 4))]
        return m

    def blob(c, r, subdiv):
        tris = icosphere(subdiv) * r * rng.uniform(0.7, 1.3, 3) + c
        return [[[_fmt(float(x)) for x in v] for v in t] for t in tris]

    def leaf(c, r):
        k = rng.random()
        o = rng.uniform(-0.5, 0.5, 3) * r
        if k < 0.45:
            return {"type": "sphere", "position": [_fmt(float(v)) for v in c + o], "radius": _fmt(float(rng.uniform(0.4, 0.9) * r))}
        if k < 0.9:
            h = rng.uniform(0.3, 0.8, 3) * r
            return {"type": "box", "point1": [_fmt(float(v)) for v in c + o - h], "point2": [_fmt(float(v)) for v in c + o + h]}
        return {"type": "mesh", "triangles": blob(c + o, 0.7 * r, 1)}

    def tree(leaves, c, r):
        if leaves == 1:
            return leaf(c, r)
        left = int(rng.integers(1, leaves))
        return {"operation": str(rng.choice(["union", "intersection", "difference"])), "first": tree(left, c, r), "second": tree(leaves - left, c, r)}

    objects = [{"type": "plane", "position": [0, -1.2, 0], "normal": [0, 1, 0], "material": material()}]
    if rng.random() < 0.5:
        objects.append({"type": "plane", "position": [0, 0, -14], "normal": [_fmt(float(rng.uniform(-0.3, 0.3))), 0, 1], "material": material()})
    for _ in range(n_objects):
        c = np.array([rng.uniform(-4, 4), rng.uniform(-0.9, 1.8), rng.uniform(-11, -3)])
        r = float(rng.uniform(0.25, 0.8))
        k = rng.random()
        if k < 0.25:
            o = {"type": "sphere", "position": [_fmt(float(v)) for v in c], "radius": _fmt(r)}
        elif k < 0.4:
            h = rng.uniform(0.4, 1.0, 3) * r
            o = {"type": "box", "point1": [_fmt(float(v)) for v in c - h], "point2": [_fmt(float(v)) for v in c + h]}
        elif k < 0.6:
            o = {"type": "mesh", "triangles": blob(c, r, int(rng.integers(0, 3)))}
        elif k < 0.68:
            tris = blob(c, r, 1)
            uv = [[[_fmt(float(u)) for u in rng.uniform(0, 1, 2)] for _ in range(3)] for _ in tris]
            o = {"type": "texturemesh", "triangles": tris, "uvCoords": uv, "texture": "fz_tex.bmp"}
        elif k < 0.75:
            o = {"type": "bumpsphere", "position": [_fmt(float(v)) for v in c], "radius": _fmt(r), "bumpmap": "fz_bump.bmp", "bumpdepth": _fmt(float(rng.uniform(0.1, 0.6)))}
        else:
            o = {"type": "csgobject"}
            o.update(tree(int(rng.integers(2, 9)), c, r))
        o["material"] = material()
        objects.append(o)
    lights = [{"type": "ambient", "color": vec(0.1, 0.3)},
              {"type": "directional", "color": vec(0.2, 0.6), "direction": [_fmt(float(rng.uniform(-0.6, 0.6))), -1.0, _fmt(float(rng.uniform(-0.6, 0.2)))]},
              {"type": "point", "color": vec(0.2, 0.7), "position": [_fmt(float(rng.uniform(-3, 3))), _fmt(float(rng.uniform(2.5, 5))), _fmt(float(rng.uniform(-8, 0)))]}]
    if rng.random() < 0.5:
        lights.append({"type": "spot", "color": vec(0.3, 0.8), "position": [_fmt(float(rng.uniform(-3, 3))), 4.0, -2.0],
                       "direction": [_fmt(float(rng.uniform(-0.3, 0.3))), -1.0, _fmt(float(rng.uniform(-1.0, -0.2)))], "cutoff": _fmt(float(rng.uniform(15, 50)))})
    if rng.random() < 0.3:
        lights.append({"type": "rectangular", "color": vec(0.2, 0.5), "position": [0.5, 4.5, -6.0], "normal": [0, -1, 0], "orientation": [1, 0, 0],
                       "width": 1.0, "height": 0.6, "samples": 4})
    scene = {"camera": {"field": 60, "background": vec(0.0, 0.3)}, "objects": objects, "lights": lights}
    path = os.path.join(out_dir, f"random_{seed}.json")
    with open(path, "w") as f:
        json.dump(scene, f)
    return path


def awkward_rays(rng, n):
    """Origins and directions for fuzzing the ray queries (tools/fuzz_rays.py): zero, denormal, tiny and huge direction
    components, no direction at all, origins on round coordinates (box faces of the generators), diagonals."""
    e = rng.uniform([-5, -1.5, -12], [5, 3, 2], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    kind = rng.integers(0, 10, n)
    for k in range(3):
        d[(kind == 1) & (rng.random(n) < 0.5), k] = 0.0                    # axis-parallel
    d[kind == 2] *= np.float32(1e-30)                                       # tiny
    d[kind == 3] *= np.float32(1e20)                                        # huge
    d[kind == 4, int(rng.integers(0, 3))] = np.float32(1e-42)               # a denormal component
    d[kind == 5] = 0.0                                                      # no direction at all
    e[kind == 6] = np.round(e[kind == 6] * 2) / 2                           # origins on a lattice (box faces of the generators use round numbers)
    d[kind == 7] = np.sign(d[kind == 7]) * np.float32(1.0)                  # diagonals
    d[kind == 8, 1] = -np.abs(d[kind == 8, 1])                              # towards the floor plane
    neg0 = (kind == 9)
    d[neg0, 0] = np.float32(-0.0)
    return e, d


def scale_scene(path, factor):
    """Writes a copy of a scene JSON with every length multiplied by `factor` (positions, radii, boxes, triangles, area-light
    sizes) and returns its path: the same picture from a camera scaled alike, with the reference's absolute epsilons
    (1e-5 acceptance thresholds, offsets of secondary rays) now large or small against the geometry."""
    with open(path) as f:
        scene = json.load(f)

    def walk(o):
        if isinstance(o, dict):
            for k, v in list(o.items()):
                if k in ("position", "point1", "point2", "triangles"):
                    o[k] = mul(v)
                elif k in ("radius", "width", "height") and isinstance(v, (int, float)):
                    o[k] = v * factor
                else:
                    walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)

    def mul(v):
        return [mul(x) for x in v] if isinstance(v, list) else v * factor

    walk(scene.get("objects", []))
    walk(scene.get("lights", []))
    out = path[:-5] + f"_x{factor:g}.json"
    with open(out, "w") as f:
        json.dump(scene, f)
    return out


def c4_scene(out_dir, samples=64):
    """BASELINE config 4: a room with CSG furniture lit by a rectangular and a circular area light, `samples` each."""
    os.makedirs(out_dir, exist_ok=True)
    objects = [
        {"type": "plane", "position": [0, -1, 0], "normal": [0, 1, 0], "material": {"diffuse": [0.55, 0.5, 0.45]}},
        {"type": "plane", "position": [0, 2.5, 0], "normal": [0, -1, 0], "material": {"diffuse": [0.6, 0.6, 0.6]}},
        {"type": "plane", "position": [-3, 0, 0], "normal": [1, 0, 0.2], "material": {"diffuse": [0.7, 0.4, 0.4]}},
        {"type": "plane", "position": [3, 0, 0], "normal": [-1, 0, 0.2], "material": {"diffuse": [0.4, 0.4, 0.7]}},
        {"type": "plane", "position": [0, 0, -8], "normal": [0, 0, 1], "material": {"diffuse": [0.6, 0.6, 0.55]}},
        {"type": "csgobject", "material": {"diffuse": [0.5, 0.3, 0.15], "specular": [0.2, 0.2, 0.2], "shininess": 10.0},
         "operation": "difference",
         "first": {"type": "box", "point1": [-1.0, -1.0, -3.5], "point2": [1.0, -0.2, -5.0]},
         "second": {"operation": "union",
                    "first": {"type": "box", "point1": [-0.8, -1.1, -3.3], "point2": [0.8, -0.35, -5.2]},
                    "second": {"type": "box", "point1": [-1.1, -1.1, -3.7], "point2": [1.1, -0.35, -4.8]}}},
        {"type": "sphere", "radius": 0.35, "position": [0.3, 0.15, -4.2],
         "material": {"diffuse": [0.9, 0.1, 0.1], "specular": [1, 1, 1], "shininess": 30.0}},
        {"type": "sphere", "radius": 0.5, "position": [-1.6, -0.5, -5.5],
         "material": {"diffuse": [0.2, 0.7, 0.3], "specular": [0.5, 0.5, 0.5], "shininess": 15.0}},
        {"type": "box", "point1": [1.2, -1.0, -5.5], "point2": [2.0, 0.4, -6.3], "material": {"diffuse": [0.7, 0.7, 0.2]}},
        {"type": "csgobject", "material": {"diffuse": [0.8, 0.5, 0.7]}, "operation": "difference",
         "first": {"type": "sphere", "position": [0, 1.6, -4.5], "radius": 0.5},
         "second": {"type": "box", "point1": [-0.6, 1.0, -3.9], "point2": [0.6, 1.6, -5.1]}},
    ]
    lights = [
        {"type": "ambient", "color": [0.15, 0.15, 0.15]},
        {"type": "rectangular", "color": [0.6, 0.6, 0.55], "position": [-1.0, 2.4, -4.0], "normal": [0, -1, 0], "orientation": [1, 0, 0],
         "width": 1.2, "height": 0.8, "samples": samples},
        {"type": "circular", "color": [0.5, 0.5, 0.55], "position": [1.5, 2.3, -5.5], "normal": [-0.2, -1, 0.1], "radius": 0.5, "samples": samples},
    ]
    scene = {"camera": {"field": 60, "background": [0, 0, 0]}, "objects": objects, "lights": lights}
    path = os.path.join(out_dir, f"c4_{samples}.json")
    with open(path, "w") as f:
        json.dump(scene, f)
    return path


def c5_scene(out_dir, grid=1581, binary=None):
    """BASELINE config 5: height field of grid*grid*2 triangles (1581 -> 4 999 122) + floor + two lights.
    As JSON text the mesh is ~0.6 GB at full size; `binary` (default: above 400 k triangles) writes it as a mesh file."""
    os.makedirs(out_dir, exist_ok=True)
    tris, _ = height_field(grid, seed=2)
    if binary is None:
        binary = len(tris) > 400000
    mesh = {"type": "mesh", "material": {"ambient": [0.1, 0.12, 0.1], "diffuse": [0.45, 0.6, 0.4], "specular": [0.2, 0.2, 0.2], "shininess": 8.0}}
    mesh.update(_mesh_file(out_dir, f"c5_{grid}.bin", tris) if binary else _mesh_json(tris))
    scene = {"camera": {"field": 60, "background": [0.3, 0.4, 0.6]},
             "objects": [mesh, {"type": "plane", "position": [0, -1.6, 0], "normal": [0, 1, 0], "material": {"diffuse": [0.3, 0.3, 0.35]}}],
             "lights": [{"type": "ambient", "color": [0.25, 0.25, 0.25]},
                        {"type": "directional", "color": [0.7, 0.7, 0.65], "direction": [-0.4, -1.0, -0.3]},
                        {"type": "point", "color": [0.5, 0.45, 0.4], "position": [2.0, 4.0, -3.0]}]}
    path = os.path.join(out_dir, f"c5_{grid}{'_bin' if binary else ''}.json")
    with open(path, "w") as f:
        json.dump(scene, f)
    return path


def icosphere(subdiv, centre=(0.0, 0.0, 0.0), radius=1.0):
    """Triangles [20 * 4^subdiv, 3, 3] of a subdivided icosahedron, wound so that normals point outwards."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    tris = np.array([[v[a], v[b], v[c]] for a, b, c in f])
    for _ in range(subdiv):
        a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        tris = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    tris = tris / np.linalg.norm(tris, axis=2, keepdims=True)
    n = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 1])
    flip = (n * tris.mean(axis=1)).sum(axis=1) < 0
    tris[flip] = tris[flip][:, ::-1]
    return (tris * radius + np.array(centre)).astype(np.float32)


def mesh_scene(out_dir, subdiv=3):
    """Meshes that are asked for hits on their own (Mesh::rayhit): a refractive icosphere (20 * 4^subdiv triangles) and a
    CSG difference whose first operand is a mesh -- the cases SURVEY.md section 8f-3 names."""
    os.makedirs(out_dir, exist_ok=True)
    glass = {"type": "mesh", "material": {"ambient": [0.05, 0.05, 0.08], "diffuse": [0.2, 0.2, 0.3], "specular": [0.6, 0.6, 0.6], "shininess": 40.0,
                                          "transmissive": [0.7, 0.7, 0.7], "refraction": 1.3}}
    glass.update(_mesh_json(icosphere(subdiv, (-0.9, 0.0, -4.0), 0.8)))
    csg_mesh = {"type": "mesh"}
    csg_mesh.update(_mesh_json(icosphere(max(subdiv - 1, 0), (1.0, -0.1, -4.5), 0.8)))
    carved = {"type": "csgobject", "material": {"ambient": [0.1, 0.05, 0.05], "diffuse": [0.8, 0.4, 0.3], "specular": [0.3, 0.3, 0.3], "shininess": 15.0},
              "operation": "difference", "first": csg_mesh, "second": {"type": "sphere", "position": [1.3, 0.2, -3.9], "radius": 0.55}}
    scene = {"camera": {"field": 60, "background": [0.1, 0.15, 0.25]},
             "objects": [glass, carved,
                         {"type": "plane", "position": [0, -1.0, 0], "normal": [0, 1, 0], "material": {"ambient": [0.1, 0.1, 0.1], "diffuse": [0.5, 0.5, 0.45]}},
                         {"type": "sphere", "radius": 0.4, "position": [-0.6, -0.3, -6.0], "material": {"ambient": [0.1, 0.1, 0.0], "diffuse": [0.9, 0.8, 0.2]}}],
             "lights": [{"type": "ambient", "color": [0.3, 0.3, 0.3]},
                        {"type": "point", "color": [0.8, 0.8, 0.75], "position": [2.0, 4.0, -1.0]},
                        {"type": "directional", "color": [0.3, 0.3, 0.35], "direction": [0.3, -1.0, -0.4]}]}
    path = os.path.join(out_dir, f"mesh_{subdiv}.json")
    with open(path, "w") as f:
        json.dump(scene, f)
    return path
