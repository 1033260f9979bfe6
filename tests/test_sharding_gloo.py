"""N > 1 path on CPU: two gloo ranks shard a frame by interleaved tiles, render their share with the oracle (the checker
stands in for the GPU here: this test is about the sharding and the gather, not the kernels), gather on rank 0 and
must reproduce the single-process frame exactly."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["RT_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RT_ROOT"], "tests"))
import raytracer_amd as rt
from raytracer_amd import distributed as rd
import oracle

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
W, H = int(os.environ["RT_W"]), int(os.environ["RT_H"])
flat = rt.FlatScene.load(os.environ["RT_SCENE"])
cam = rt.Camera(flat.fov, W, H)
# this rank's pixels, in the device's tile order
pix = rd.tile_pixels(W, H, rank, world)
ok = pix >= 0
targets = oracle.primary_targets(cam, W, H)
r = oracle.trace_batch(flat, np.zeros((ok.sum(), 3), np.float32), targets[pix[ok]], seed=1)
packed = torch.zeros(len(pix), 3)
packed[torch.from_numpy(ok)] = torch.from_numpy(r["rgb"])
sizes = [rd.tiles_count(W, H, q, world) * 256 * 3 for q in range(world)]
bufs = rd.gather_packed(packed.reshape(-1), sizes, rank, world)
if rank == 0:
    frame = torch.zeros(H, W, 3)
    for q in range(world):
        rd.unpack_tiles(bufs[q].reshape(-1, 3), frame, q, world)
    ref, _, _ = oracle.render(flat, cam, W, H, seed=1)
    assert np.array_equal(frame.numpy().view(np.uint32), ref.view(np.uint32)), "sharded frame differs from the single-process frame"
    print("SHARDED_OK")
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,W,H", [(2, 160, 120), (3, 100, 52)])   # 3 ranks / ragged tile counts and partial edge tiles
def test_two_rank_tile_sharding_and_gather(world, W, H, scene_dir, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RT_ROOT=ROOT, RT_SCENE=str(scene_dir / "c.rtscene"), RT_W=str(W), RT_H=str(H), MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(29500 + world), str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "SHARDED_OK" in out.stdout


def test_tile_map_is_a_partition():
    from raytracer_amd import distributed as rd
    for W, H, ranks in [(160, 120, 2), (200, 100, 3), (1920, 1080, 8), (33, 9, 4)]:
        seen = np.zeros(W * H, np.int32)
        for r in range(ranks):
            pix = rd.tile_pixels(W, H, r, ranks)
            assert len(pix) == rd.tiles_count(W, H, r, ranks) * 256
            np.add.at(seen, pix[pix >= 0], 1)
        assert (seen == 1).all()
