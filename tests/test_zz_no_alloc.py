"""GPU suite, last part: rt_intersect_batch_static -- the entry point that allocates nothing (its queues and counters are a
file-scope device array of the library) -- gives exactly what rt_intersect_batch gives: hit mask, object ids, t bit for bit.
Two calls in a row on one scene, the second with a smaller input (the array keeps the first call's contents), then another
scene on the same GPU (the array is per GPU, not per handle), the size limit, and the empty batch."""
import numpy as np
import pytest
import torch

import raytracer_amd as rt
from raytracer_amd import capi
from raytracer_amd.capi import RtError

pytestmark = pytest.mark.gpu


def rays_for(flat, n, seed):
    """Rays from around the camera position into the scene's bounding box, some of them awkward (axis-parallel, pointing away)."""
    rng = np.random.default_rng(seed)
    box = np.array(flat.info.root_box[:], np.float32).reshape(3, 2)
    lo, hi = box[:, 0], box[:, 1]
    if not np.all(np.isfinite(box)) or np.any(hi < lo):
        lo, hi = np.full(3, -5, np.float32), np.full(3, 5, np.float32)
    e = (rng.normal(size=(n, 3)) * 0.5).astype(np.float32)
    target = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = target - e
    d[::17, 0] = 0.0
    d[5::23] *= -1.0
    return np.ascontiguousarray(e), np.ascontiguousarray(d)


def static_call(sc, e, d):
    n = len(e)
    de, dd = torch.from_numpy(e).cuda(), torch.from_numpy(d).cuda()
    # outputs start from garbage: the call must write every slot
    hit = torch.full((max(n, 1),), 7, dtype=torch.uint8, device="cuda")
    oid = torch.full((max(n, 1),), 12345, dtype=torch.int32, device="cuda")
    t = torch.full((max(n, 1),), -3.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    sc.intersect_batch_static(de.data_ptr(), dd.data_ptr(), n, hit.data_ptr(), oid.data_ptr(), t.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    return hit.cpu().numpy()[:n], oid.cpu().numpy()[:n], t.cpu().numpy()[:n]


def same(sc, e, d):
    ref = sc.intersect_batch(e, d)
    hit, oid, t = static_call(sc, e, d)
    assert np.array_equal(hit, ref["hit"])
    assert np.array_equal(oid, ref["id"])
    assert np.array_equal(t.view(np.uint32), ref["t"].view(np.uint32))
    return int(ref["hit"].sum())


def test_static_entry_point_equals_intersect_batch(gpu_scenes, flat_scenes):
    sc, flat = gpu_scenes("i"), flat_scenes("i")          # 242-triangle meshes + spheres + planes
    e1, d1 = rays_for(flat, 50000, 1)
    e2, d2 = rays_for(flat, 3001, 2)                       # smaller, not a multiple of anything
    assert same(sc, e1, d1) > 1000
    assert same(sc, e2, d2) > 100
    # another handle on the same GPU shares the array (CSG objects: scene k)
    sk, fk = gpu_scenes("k"), flat_scenes("k")
    e3, d3 = rays_for(fk, 20000, 3)
    assert same(sk, e3, d3) > 100
    assert same(sc, e2, d2) > 100
    # the full capacity, and one ray
    e4, d4 = rays_for(flat, capi.RT_STATIC_MAX_RAYS, 4)
    assert same(sc, e4, d4) > 1000
    assert same(sc, e4[:1], d4[:1]) >= 0


def test_static_entry_point_limits(gpu_scenes, flat_scenes):
    sc, flat = gpu_scenes("c"), flat_scenes("c")
    e, d = rays_for(flat, capi.RT_STATIC_MAX_RAYS + 1, 5)
    de, dd = torch.from_numpy(e).cuda(), torch.from_numpy(d).cuda()
    out = torch.zeros(len(e), dtype=torch.int32, device="cuda")
    with pytest.raises(RtError) as err:
        sc.intersect_batch_static(de.data_ptr(), dd.data_ptr(), len(e), 0, out.data_ptr(), 0)
    assert err.value.code == capi.RT_ELIMIT and "RT_STATIC_MAX_RAYS" in str(err.value)
    sc.intersect_batch_static(0, 0, 0, 0, 0, 0)            # nothing to do is not an error
    with pytest.raises(RtError):
        sc.intersect_batch_static(0, dd.data_ptr(), 4, 0, out.data_ptr(), 0)
    # outputs are optional
    sc.intersect_batch_static(de.data_ptr(), dd.data_ptr(), 1000, 0, out.data_ptr(), 0)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy()[:1000], sc.intersect_batch(e[:1000], d[:1000])["id"])
