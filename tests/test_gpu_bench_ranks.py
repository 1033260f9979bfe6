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
           "--width", "960", "--height", "540", "--grid", "96"]
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
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0 and out["cpu_baseline"] is None
