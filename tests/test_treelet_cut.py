"""rt_scene_upload cuts the tree into treelets on several threads (cutTree, rt_capi_device.hip): the top of the tree sequentially, the
subtrees below it independently, their references moved to where they land.  The treelets' numbers change, nothing else may: this test
reduces both forms (binary treelets, quad treelets) to a digest of the tree as a walk from the root sees it -- boxes, aux words, embedded
triangles, leaf references, the slot and the record every reference leads to -- and compares the cut on 1 thread with cuts on 2, 5 and
16, on trees from a few hundred to 435 600 triangles (fat leaves with sub-trees included).  Host code only: rt_debug_treelet_digest."""
import ctypes as C
import gzip
import os
import shutil

import pytest

import raytracer_amd as rt
from raytracer_amd import capi, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest(flat, threads):
    lib = capi.load()
    fn = lib.rt_debug_treelet_digest
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 4)()
    rc = fn(C.cast(flat._h, C.c_void_p), threads, out)
    assert rc == 0, lib.rt_last_error()
    return tuple(out)


def flats(tmp):
    yield "C2 100 352 triangles", rt.FlatScene.from_json(scenes.c2_scene(str(tmp / "c2"), grid=224))
    yield "C5 435 600 triangles (fat leaves)", rt.FlatScene.from_json(scenes.c5_scene(str(tmp / "c5"), 467, binary=True))
    yield "C5 20 000 triangles", rt.FlatScene.from_json(scenes.c5_scene(str(tmp / "c5s"), 100, binary=True))
    for ch in "jl":
        dst = str(tmp / f"{ch}.rtscene")
        with gzip.open(os.path.join(ROOT, "tests", "golden", "scenes", f"{ch}.rtscene.gz"), "rb") as a, open(dst, "wb") as b:
            shutil.copyfileobj(a, b)
        yield f"scene {ch}", rt.FlatScene.load(dst)


def test_parallel_cut_is_the_sequential_cut_renumbered(tmp_path):
    for name, flat in flats(tmp_path):
        want = digest(flat, 1)
        assert want[2] > 0 and want[3] > 0, name
        for threads in (2, 5, 16):
            assert digest(flat, threads) == want, (name, threads)
