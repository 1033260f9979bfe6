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
    # free device memory as the DRIVER counts it (hipMemGetInfo): torch's own statistics only see torch's caching allocator, a hipMalloc
    # inside the library would pass them unnoticed.  (The first call on a handle loads its kernels, which may take memory: the
    # comparison is made from the second call on; the graph-capture test below shows the same thing another way.)
    warm = getattr(sc, "_static_warm", False)
    before = torch.cuda.mem_get_info()[0]
    sc.intersect_batch_static(de.data_ptr(), dd.data_ptr(), n, hit.data_ptr(), oid.data_ptr(), t.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    after = torch.cuda.mem_get_info()[0]
    if warm:
        assert after == before, f"free device memory went from {before} to {after} across the call"
    sc._static_warm = True
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


def test_static_entry_point_under_graph_capture(gpu_scenes, flat_scenes):
    """The call captured into a HIP graph (torch.cuda.graph: global capture mode, in which an allocating or synchronising runtime call
    made by the capturing thread fails the capture), replayed twice -- the second time on other rays written into the same buffers --
    and compared with rt_intersect_batch.  A capture with no ordinary call before it on the handle is refused."""
    import raytracer_amd as rt
    flat = flat_scenes("i")
    sc = rt.Scene(flat, 0)                                  # a fresh handle: nothing has run on it yet
    n = 20000
    e1, d1 = rays_for(flat, n, 11)
    e2, d2 = rays_for(flat, n, 12)
    de, dd = torch.from_numpy(e1).cuda(), torch.from_numpy(d1).cuda()
    hit = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
    oid = torch.full((n,), 12345, dtype=torch.int32, device="cuda")
    t = torch.full((n,), -3.0, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    # (1) cold handle: refused, and the refusal leaves the capture intact (nothing was recorded, nothing illegal was called)
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0, stream=side):
        with pytest.raises(RtError) as err:
            sc.intersect_batch_static(de.data_ptr(), dd.data_ptr(), n, hit.data_ptr(), oid.data_ptr(), t.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        hit.add_(0)                                         # (an empty capture is not worth instantiating)
    assert err.value.code == capi.RT_EINVAL and "outside the capture" in str(err.value)
    # (2) one ordinary call, then the capture
    sc.intersect_batch_static(de.data_ptr(), dd.data_ptr(), n, hit.data_ptr(), oid.data_ptr(), t.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    hit.fill_(7); oid.fill_(12345); t.fill_(-3.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        sc.intersect_batch_static(de.data_ptr(), dd.data_ptr(), n, hit.data_ptr(), oid.data_ptr(), t.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(hit[0]) == 7 and int(oid[0]) == 12345        # captured, not executed
    for e, d in ((e1, d1), (e2, d2)):
        de.copy_(torch.from_numpy(e)); dd.copy_(torch.from_numpy(d))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ref = sc.intersect_batch(e, d)
        assert np.array_equal(hit.cpu().numpy(), ref["hit"]) and np.array_equal(oid.cpu().numpy(), ref["id"])
        assert np.array_equal(t.cpu().numpy().view(np.uint32), ref["t"].view(np.uint32))
        assert int(ref["hit"].sum()) > 100
    del g, g0
    sc.close()
