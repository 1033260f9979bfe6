"""GPU suite, part 3: frames in flight (rt_render_submit / rt_render_wait) give exactly the frames of the synchronous call,
whatever the camera changes between them, and the ticket rules are enforced."""
import numpy as np
import pytest
import torch

import raytracer_amd as rt
from raytracer_amd import scenes
from raytracer_amd.capi import RtError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2(tmp_path_factory):
    flat = rt.FlatScene.from_json(scenes.c2_scene(str(tmp_path_factory.mktemp("c2async"), ), grid=96))
    sc = rt.Scene(flat, 0)
    yield flat, sc
    sc.close()


def test_frames_in_flight_equal_synchronous_frames(c2):
    flat, sc = c2
    W, H = 960, 540
    cams = [rt.Camera(flat.fov, W, H, eye=(0.1 * i, 1.5, 1.0 - 0.2 * i), rot_x=-0.3, rot_y=0.05 * i) for i in range(6)]
    ref = []
    buf = torch.zeros(H, W, 3, device="cuda")
    for cam in cams:
        sc.render_into(buf.data_ptr(), W, H, camera=cam)
        ref.append(buf.cpu().numpy().copy())
    bufs = [torch.zeros(H, W, 3, device="cuda") for _ in range(3)]
    pending, got, rays = [], [], []
    for i, cam in enumerate(cams):
        if len(pending) == 3:
            t, b = pending.pop(0)
            rays.append(sc.render_wait(t)["rays_nearest"])
            got.append(b.cpu().numpy().copy())
        b = bufs[i % 3]
        pending.append((sc.render_submit(b.data_ptr(), W, H, camera=cam), b))
    for t, b in pending:
        rays.append(sc.render_wait(t)["rays_nearest"])
        got.append(b.cpu().numpy().copy())
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert np.array_equal(g.view(np.uint32), r.view(np.uint32))
    assert all(n == W * H for n in rays)


def test_mixed_sizes_and_other_entry_points_between_frames(c2):
    flat, sc = c2
    a, b = torch.zeros(270, 480, 3, device="cuda"), torch.zeros(540, 960, 3, device="cuda")
    cam_a, cam_b = rt.Camera(flat.fov, 480, 270, eye=(0, 1.5, 1.0), rot_x=-0.3), rt.Camera(flat.fov, 960, 540, eye=(0, 1.5, 1.0), rot_x=-0.3)
    ta = sc.render_submit(a.data_ptr(), 480, 270, camera=cam_a)
    tb = sc.render_submit(b.data_ptr(), 960, 540, camera=cam_b)          # another size: the lanes run dry first
    e = np.zeros((64, 3), np.float32) + np.array([0, 1.5, 1.0], np.float32)
    s = e + np.array([0, -0.3, -1.0], np.float32)
    hits = sc.trace_batch(e, s)                                           # waits for the frames in flight
    sc.render_wait(tb)
    sc.render_wait(ta)
    ra, rb = torch.zeros_like(a), torch.zeros_like(b)
    sc.render_into(ra.data_ptr(), 480, 270, camera=cam_a)
    sc.render_into(rb.data_ptr(), 960, 540, camera=cam_b)
    assert torch.equal(a, ra) and torch.equal(b, rb)
    assert hits["hit"].all()


def test_ticket_rules(c2):
    flat, sc = c2
    W, H = 480, 270
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    bufs = [torch.zeros(H, W, 3, device="cuda") for _ in range(9)]
    tickets = [sc.render_submit(bufs[i].data_ptr(), W, H, camera=cam) for i in range(8)]
    with pytest.raises(RtError):
        sc.render_submit(bufs[8].data_ptr(), W, H, camera=cam)            # a ninth frame in flight
    for t in tickets:
        sc.render_wait(t)
    with pytest.raises(RtError):
        sc.render_wait(tickets[0])                                        # waited for already
    with pytest.raises(RtError):
        sc.render_wait(12345)
    assert all(torch.equal(bufs[0], b) for b in bufs[1:8])


def test_handle_freed_with_frames_in_flight(c2):
    """rt_scene_free with submitted, un-waited frames: the lanes finish them before anything they use is released."""
    flat, _ = c2
    W, H = 960, 540
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    for _ in range(3):
        sc = rt.Scene(flat, 0)
        bufs = [torch.zeros(H, W, 3, device="cuda") for _ in range(3)]
        for b in bufs:
            sc.render_submit(b.data_ptr(), W, H, camera=cam)
        sc.close()
        torch.cuda.synchronize()
        assert torch.equal(bufs[0], bufs[1]) and torch.equal(bufs[0], bufs[2]) and float(bufs[0].abs().sum()) > 0


def test_tickets_waited_out_of_order_free_their_slots(c2):
    """Up to 8 tickets outstanding, waited for in any order: a slot is free as soon as ITS ticket was waited for."""
    flat, sc = c2
    W, H = 480, 270
    cam = rt.Camera(flat.fov, W, H, eye=(0, 1.5, 1.0), rot_x=-0.3)
    bufs = [torch.zeros(H, W, 3, device="cuda") for _ in range(8)]
    t = [sc.render_submit(bufs[i].data_ptr(), W, H, camera=cam) for i in range(8)]
    for k in range(1, 8):
        sc.render_wait(t[k])
    more = [sc.render_submit(bufs[k].data_ptr(), W, H, camera=cam) for k in range(1, 8)]   # ticket t[0] is still outstanding
    assert len(set(more + t)) == 15
    sc.render_wait(t[0])
    for m in reversed(more):
        sc.render_wait(m)
    assert all(torch.equal(bufs[0], b) for b in bufs[1:])


def test_other_entry_points_wait_for_the_gpu_not_only_for_the_lanes(c2):
    """A large frame, then a smaller one, neither waited for, then trace_batch: the batch uses lane 0's queues and the shared
    accumulators, so it must wait until the GPU is through with the frames (not just until they are enqueued)."""
    flat, sc = c2
    big, small = torch.zeros(540, 960, 3, device="cuda"), torch.zeros(270, 480, 3, device="cuda")
    cam_b, cam_s = rt.Camera(flat.fov, 960, 540, eye=(0, 1.5, 1.0), rot_x=-0.3), rt.Camera(flat.fov, 480, 270, eye=(0.2, 1.4, 1.0), rot_x=-0.25)
    rng = np.random.default_rng(3)
    e = (np.array([0, 1.5, 1.0], np.float32) + 0.05 * rng.normal(size=(4096, 3))).astype(np.float32)
    s = (e + np.array([0, -0.3, -1.0], np.float32) + 0.2 * rng.normal(size=(4096, 3))).astype(np.float32)
    want = sc.trace_batch(e, s)
    rb, rs = torch.zeros_like(big), torch.zeros_like(small)
    sc.render_into(rb.data_ptr(), 960, 540, camera=cam_b)
    sc.render_into(rs.data_ptr(), 480, 270, camera=cam_s)
    for _ in range(3):
        big.zero_(); small.zero_()
        tb = sc.render_submit(big.data_ptr(), 960, 540, camera=cam_b)
        ts = sc.render_submit(small.data_ptr(), 480, 270, camera=cam_s)
        got = sc.trace_batch(e, s)
        sc.render_wait(ts)
        sc.render_wait(tb)
        assert torch.equal(big, rb) and torch.equal(small, rs)
        for k in ("hit", "id"):
            assert np.array_equal(got[k], want[k])
        assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
        assert np.array_equal(got["rgb"].view(np.uint32), want["rgb"].view(np.uint32))


@pytest.mark.parametrize("aa", [1, 2])
def test_whole_frames_on_one_lane_equal_sharded_frames(c2, aa):
    """RT_FLAG_WHOLE_FRAME: a frame in flight is rendered on ONE lane (its own accumulators), frames dealt to the lanes in turn, instead of
    as a shard per lane.  Same pixels bit for bit and the same ray counts -- whole and sharded frames mixed in flight, cameras changing,
    a tile split as a rank of a multi-GPU job would ask for, and a frame size change in the middle (the lanes' accumulators grow)."""
    flat, sc = c2
    jobs = []
    for i in range(9):
        W, H = (960, 540) if i < 6 else (640, 360)
        cam = rt.Camera(flat.fov, W, H, eye=(0.1 * i, 1.5, 1.0 - 0.2 * i), rot_x=-0.3, rot_y=0.05 * i)
        jobs.append((W, H, cam, (i % 3, 3) if i in (4, 5) else (0, 1)))
    ref = []
    for W, H, cam, (tr, trs) in jobs:
        buf = torch.zeros(H, W, 3, device="cuda")
        st = sc.render_into(buf.data_ptr(), W, H, camera=cam, aa=aa, tile_rank=tr, tile_ranks=trs)
        ref.append((buf.cpu().numpy().copy(), st["rays_nearest"], st["rays_shadow"]))
    bufs = [torch.zeros(540, 960, 3, device="cuda") for _ in range(3)]
    pending, got = [], []

    def retire():
        t, b, W, H = pending.pop(0)
        st = sc.render_wait(t)
        got.append((b.view(-1)[:H * W * 3].view(H, W, 3).cpu().numpy().copy(), st["rays_nearest"], st["rays_shadow"]))
    for i, (W, H, cam, (tr, trs)) in enumerate(jobs):
        if len(pending) == 3:
            retire()
        b = bufs[i % 3]
        b.zero_()
        flags = 0 if i == 2 else rt.RT_FLAG_WHOLE_FRAME          # (one sharded frame in the middle of the whole ones)
        pending.append((sc.render_submit(b.view(-1)[:H * W * 3].data_ptr(), W, H, camera=cam, aa=aa, tile_rank=tr, tile_ranks=trs, flags=flags), b, W, H))
    while pending:
        retire()
    for (a, an, ash), (b, bn, bsh) in zip(ref, got):
        assert (an, ash) == (bn, bsh)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_queues_grow_for_a_large_frame(c2, monkeypatch):
    """A frame that brings more rays per lane than two batches of the present queues hold makes rt_render_submit grow them (8 Mi -> up to
    32 Mi entries; RT_QUEUE_RAYS fixes the size): frames already in flight are finished first, and the large frame -- now a few large
    batches per lane instead of many -- has the same pixels, bit for bit, and the same ray counts as with the fixed queues."""
    flat, _ = c2
    W, H, aa = 1920, 1080, 4                                               # 33 M primary rays, 8.3 M per lane
    cam = rt.Camera(flat.fov, W, H, eye=(0.1, 1.5, 1.0), rot_x=-0.3)
    small_cam = rt.Camera(flat.fov, 480, 270, eye=(0, 1.5, 1.0), rot_x=-0.3)
    monkeypatch.setenv("RT_QUEUE_RAYS", str(8 << 20))
    fixed = rt.Scene(flat, 0)
    monkeypatch.delenv("RT_QUEUE_RAYS")
    ref, ref_small = torch.zeros(H, W, 3, device="cuda"), torch.zeros(270, 480, 3, device="cuda")
    st_ref = fixed.render_into(ref.data_ptr(), W, H, camera=cam, aa=aa)
    fixed.render_into(ref_small.data_ptr(), 480, 270, camera=small_cam)
    fixed.close()
    sc = rt.Scene(flat, 0)
    big, small = torch.zeros(H, W, 3, device="cuda"), torch.zeros(270, 480, 3, device="cuda")
    ts = sc.render_submit(small.data_ptr(), 480, 270, camera=small_cam)   # in flight while the queues grow
    tb = sc.render_submit(big.data_ptr(), W, H, camera=cam, aa=aa)
    st_big = sc.render_wait(tb)
    sc.render_wait(ts)
    assert torch.equal(big, ref) and torch.equal(small, ref_small)
    assert (st_big["rays_nearest"], st_big["rays_shadow"]) == (st_ref["rays_nearest"], st_ref["rays_shadow"])
    assert st_big["rays_nearest"] == W * H * aa * aa
    again = torch.zeros(H, W, 3, device="cuda")
    sc.render_into(again.data_ptr(), W, H, camera=cam, aa=aa, flags=rt.RT_FLAG_WHOLE_FRAME)
    small.zero_()
    sc.render_into(small.data_ptr(), 480, 270, camera=small_cam)
    assert torch.equal(again, ref) and torch.equal(small, ref_small)
    sc.close()


def test_queue_growth_falls_back_when_device_memory_is_short(c2):
    """The device cannot give the larger queues (here: a tensor holds all but 3 GB of its memory): the growth is halved down to the size the
    queues had, whose memory was just released, and the large frame is rendered in more batches -- same pixels, same counts."""
    flat, _ = c2
    W, H, aa = 1920, 1080, 4
    cam = rt.Camera(flat.fov, W, H, eye=(0.1, 1.5, 1.0), rot_x=-0.3)
    sc = rt.Scene(flat, 0)
    ref, out = torch.zeros(H, W, 3, device="cuda"), torch.zeros(H, W, 3, device="cuda")
    small = torch.zeros(270, 480, 3, device="cuda")
    sc.render_into(small.data_ptr(), 480, 270, camera=rt.Camera(flat.fov, 480, 270, eye=(0, 1.5, 1.0), rot_x=-0.3))   # (accumulators, module load)
    torch.cuda.synchronize()
    free, _ = torch.cuda.mem_get_info()
    try:
        hog = torch.empty(int(free - 3 * 2 ** 30), dtype=torch.uint8, device="cuda")
    except RuntimeError:                                                   # (someone else on the card, or a fragmented pool: nothing to test with)
        sc.close()
        pytest.skip("cannot take the device's free memory in one piece")
    try:
        st = sc.render_into(out.data_ptr(), W, H, camera=cam, aa=aa)       # wants 16 Mi-entry queues: 12 GB where 6 + 3 are to be had
    finally:
        del hog
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    before, _ = torch.cuda.mem_get_info()
    st_ref = sc.render_into(ref.data_ptr(), W, H, camera=cam, aa=aa)       # (now they grow: 6 -> 12 GB)
    torch.cuda.synchronize()
    after, _ = torch.cuda.mem_get_info()
    assert before - after > 4 * 2 ** 30                                    # ... so they had not grown under the hog
    assert torch.equal(out, ref)
    assert (st["rays_nearest"], st["rays_shadow"]) == (st_ref["rays_nearest"], st_ref["rays_shadow"]) and st["rays_nearest"] == W * H * aa * aa
    sc.close()
