"""GPU suite, part 4: bench.py with more than one rank.  A one-GPU box cannot host an RCCL group of two, so the ranks share
device 0 and exchange their tiles through the host over gloo (RT_BENCH_REHEARSAL=1): everything else -- the tile split, the
sample grid growing with the ranks, frames in flight, pack / gather / unpack on the side stream, buffer reuse -- is the code the
driver runs at N = 2, 4, 8.  The bench itself checks that the frame assembled from the ranks' tiles is, bit for bit, the frame
one rank renders alone, and reports it in its JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("ranks", [2, 3])
def test_bench_line_with_several_ranks_sharing_one_gpu(ranks):
    env = dict(os.environ, RT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "7", "--warmup", "2",
           "--width", "960", "--height", "540", "--grid", "96", "--c4-samples", "8", "--config-steps", "3"] + (["--rgb8"] if ranks == 3 else [])
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["steps"] == 7 and out["scaling"] == "weak"
    assert out["assembled_equals_single_rank"] is True
    assert out["config"]["samples_per_pixel"] == {2: 2, 3: 3}[ranks]
    # weak scaling: every rank traces one sample per pixel of its share, so the job's primary rays grow with the ranks
    assert out["config"]["rays_per_step"] >= ranks * 960 * 540
    assert out["value"] > 0 and out["roofline"]["launch_ms"] > 0 and out["cpu_baseline"] is None
    assert out["exchange"] == ("rgb8" if ranks == 3 else "f32")
    # the timed region: blocks of exactly `steps` frames repeated until half a second has gone by, the median block reported
    assert out["blocks"] >= 1 and out["ms_per_step_min"] <= out["ms_per_step"] <= out["ms_per_step_max"]
    # the same N-samples-per-pixel frame on one GPU alone, and the efficiency that follows from it (not from value(1), whose frame is another)
    assert out["one_gpu_same_workload_ms"] > 0
    assert abs(out["efficiency"] - out["one_gpu_same_workload_ms"] / (ranks * out["ms_per_step"])) < 1e-3
    # the multi-GPU configuration BASELINE.json names for this many ranks, one frame strong-scaled over them, gather inside the timed region
    c4 = out["baseline_configs"]["C4"]
    assert c4["scaling"] == "strong" and c4["n_gpus"] == ranks and c4["assembled_equals_single_rank"] is True and c4["gather_in_timed_region"] is True
    # every hit lights 2 x 8 samples: rays = primary rays + 16 shadow rays per surface, the same whatever the number of ranks
    assert c4["rays_per_frame"] > 960 * 540 and (c4["rays_per_frame"] - 960 * 540) % 16 == 0
    assert c4["value"] > 0 and c4["one_gpu_ms_per_frame"] > 0


def test_bench_c5_path_with_eight_ranks_worth_of_arguments():
    """The N >= 8 branch (C5: big mesh, 4x4 samples per pixel at twice the bench resolution per axis) on a reduced mesh with 2 ranks is not
    reachable through --gpus; its workload builder and runner are exercised directly instead, at a reduced size."""
    import argparse
    import tempfile
    sys.path.insert(0, ROOT)
    import torch
    import bench
    import raytracer_amd as rt
    from raytracer_amd import scenes
    args = argparse.Namespace(width=240, height=136, grid=24, c4_samples=8, c5_grid=60)
    with tempfile.TemporaryDirectory() as tmp:
        wl = bench.make_workload("c5", args, 8, rt, scenes, tmp)
    assert (wl.W, wl.H, wl.aa) == (480, 272, (4, 4)) and wl.flat.info.n_tris == 60 * 60 * 2
    ctx = bench.Ctx()
    ctx.rt, ctx.torch, ctx.world, ctx.rank, ctx.local_rank, ctx.rehearsal, ctx.dev = rt, torch, 1, 0, 0, False, torch.device("cuda", 0)
    ctx.barrier = lambda: torch.cuda.synchronize()
    ctx.max_over_ranks = ctx.sum_over_ranks = lambda x: x
    r = bench.Runner(ctx, wl, 3)
    elapsed, rays, latency, blocks = r.timed(4, 1)
    assert len(blocks) >= 1 and min(blocks) <= elapsed <= max(blocks)
    assert rays >= 4 * 480 * 272 * 16 and elapsed > 0 and latency > 0
    # tiles of 8 shards compose to the same frame
    whole = torch.zeros(wl.H, wl.W, 3, device="cuda")
    r.scene.render_into(whole.data_ptr(), wl.W, wl.H, aa=wl.aa, camera=r.cam)
    parts = torch.zeros_like(whole)
    for k in range(8):
        r.scene.render_into(parts.data_ptr(), wl.W, wl.H, aa=wl.aa, camera=r.cam, tile_rank=k, tile_ranks=8)
    assert torch.equal(whole.view(torch.int32), parts.view(torch.int32))
    r.close()
