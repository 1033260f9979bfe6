#!/usr/bin/env python3
"""bench.py -- Mrays/s of the MI355X trace path on BASELINE.json's 1080p workload (config 2).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one frame of the C2 scene (100 352-triangle textured height field + two bump-mapped spheres, ambient +
directional + point light) at 1920x1080: generate -> intersect -> shade -> shadow -> resolve on the GPU, scene and
queues resident in HBM, nothing crossing PCIe inside the timed region.  With N > 1 ranks the frame is sharded by
interleaved 32x8 tiles (scene replicated per GPU), each rank packs its tiles and rank 0 gathers them over RCCL and
unpacks them into the full frame -- that exchange is inside the timed region.  Scaling is WEAK: the sample grid
grows with N (1x1, 2x1, 2x2, 4x2 samples per pixel for N = 1, 2, 4, 8) so every GPU traces one sample per pixel of
its share, i.e. 1920*1080 primary rays per GPU at every N.

Mrays = BVH::findNearest calls + BVH::calcShadow calls (the reference's own two ray queries, bvh.cpp:72,123).
The JSON line also carries the roofline of the dominant kernel and the CPU restatement timed on this host.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

AA_GRID = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# algorithmic bytes per unit of traversal work (with this build's 64 B two-child node)
NODE_BYTES, TRI_BYTES, SPHERE_BYTES, RAY_QUEUE_BYTES, HIT_QUEUE_BYTES, SURF_BYTES = 64, 64, 32, 44, 12, 84


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--grid", type=int, default=224, help="height-field cells per side (224 -> 100 352 triangles)")
    ap.add_argument("--inflight", type=int, default=3, help="frames in flight (rt_render_submit / rt_render_wait); 1 = one synchronous frame at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--save", default=None, help="write the last frame as a BMP (rank 0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd import scenes
    from raytracer_amd import distributed as rd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: RT_BENCH_REHEARSAL=1 puts every rank on device 0 and gathers through the host over gloo
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}; launch N > 1 through torch.distributed.run", file=sys.stderr)
        args.gpus = world
    if rt.device_count() < 1:
        raise RuntimeError("bench.py needs a HIP device: the trace path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    W, H = args.width, args.height
    ax, ay = AA_GRID.get(world, (world, 1))
    eye, rot_x = (0.0, 1.5, 1.0), -0.3

    # scene: generated deterministically in the reference's JSON schema, loaded by the library's choose_scene restatement
    with tempfile.TemporaryDirectory() as tmp:
        path = scenes.c2_scene(tmp, grid=args.grid)
        flat = rt.FlatScene.from_json(path)
    scene = rt.Scene(flat, local_rank)
    cam = rt.Camera(flat.fov, W, H, eye=eye, rot_x=rot_x)

    D = max(1, min(3, args.inflight))
    frames = [torch.zeros(H, W, 3, device=dev, dtype=torch.float32) for _ in range(D)]
    frame = frames[0]
    stream = torch.cuda.current_stream(dev).cuda_stream
    # gather payload: every rank's buffer padded to the largest tile count (8100 tiles do not divide by 8), so that the
    # exchange is ONE grouped gather -- all peers send to the root at once, each over its own xGMI link -- instead of a
    # chain of point-to-point receives
    pad = max(scene.tiles_count(W, H, r, world) for r in range(world)) * 256 * 3
    packed = torch.zeros(pad, device=dev, dtype=torch.float32)
    sizes = [pad] * world
    recv = [torch.zeros(n, device=dev, dtype=torch.float32) for n in sizes] if (world > 1 and rank == 0) else None

    # The exchange runs on a stream of its own.  Lane 0 of the renderer works on the caller's stream (torch's current one):
    # a gather queued there would hold that lane's next shards back until every rank had arrived.  Order is kept with two
    # events per frame buffer: the side stream starts after the frame is complete (rt_render_wait has waited for it on the
    # host), and the next frame rendered into the same buffer starts after that buffer's tiles were packed and unpacked.
    side = torch.cuda.Stream(device=dev) if world > 1 else None
    assembled = [None] * D

    def assemble(slot):
        # tiles -> rank 0 (one grouped RCCL gather over xGMI, every peer sending to the root at once), then de-interleave
        buf = frames[slot]
        with torch.cuda.stream(side):
            scene.tiles_pack(buf.data_ptr(), packed.data_ptr(), W, H, rank, world, stream=side.cuda_stream)
            if rehearsal:
                host = rd.gather_packed(packed.cpu(), sizes, rank, world)
                bufs = [h.to(dev) for h in host] if rank == 0 else None
            else:
                bufs = rd.gather_packed(packed, sizes, rank, world, bufs=recv)
            if rank == 0:
                for r in range(1, world):
                    scene.tiles_unpack(bufs[r].data_ptr(), buf.data_ptr(), W, H, r, world, stream=side.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(side)
        assembled[slot] = ev

    def buffer_free(slot):
        """The caller's stream (and with it the frame about to be submitted) waits until the buffer's last exchange is through."""
        if assembled[slot] is not None:
            torch.cuda.current_stream(dev).wait_event(assembled[slot])
            assembled[slot] = None

    def step(flags=0, max_depth=rt.MAX_REFLECTIONS):
        """One synchronous frame."""
        buffer_free(0)
        st = scene.render_into(frame.data_ptr(), W, H, aa=(ax, ay), camera=cam, tile_rank=rank, tile_ranks=world, stream=stream, flags=flags,
                               max_depth=max_depth)
        if world > 1:
            assemble(0)
            side.synchronize()
        return st

    def run_steps(k):
        """k frames with up to D in flight: frame i+1 is queued before frame i is waited for (and, with N > 1 ranks, gathered)."""
        rays, pending = 0, []
        def retire():
            nonlocal rays
            ticket, slot = pending.pop(0)
            st = scene.render_wait(ticket)
            if world > 1:
                assemble(slot)
            rays += st["rays_nearest"] + st["rays_shadow"]
        for i in range(k):
            if len(pending) == D:
                retire()
            slot = i % D
            buffer_free(slot)
            pending.append((scene.render_submit(frames[slot].data_ptr(), W, H, aa=(ax, ay), camera=cam, tile_rank=rank, tile_ranks=world, stream=stream), slot))
        while pending:
            retire()
        return rays

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    rays = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    # latency of a frame rendered alone (nothing else in flight), for the record
    barrier()
    t1 = time.perf_counter()
    for _ in range(5):
        step()
    barrier()
    latency_ms = (time.perf_counter() - t1) / 5 * 1e3
    # outside the timed region: the frame assembled from every rank's tiles must be, bit for bit, the frame one rank renders alone
    assembled_ok = None
    if world > 1:
        if rank == 0:
            alone = torch.zeros(H, W, 3, device=dev, dtype=torch.float32)
            scene.render_into(alone.data_ptr(), W, H, aa=(ax, ay), camera=cam, tile_rank=0, tile_ranks=1, stream=stream)
            assembled_ok = bool(torch.equal(alone.view(torch.int32), frame.view(torch.int32)))
            del alone
        barrier()
    if world > 1:
        cdev = "cpu" if rehearsal else dev
        tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rays], device=cdev, dtype=torch.int64)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays = int(rr.item())

    # ---- roofline of the dominant kernel: live hipEvent timing of the traversal launches + an untimed counting pass ----
    # The two traversal kernels of level 0 (primary rays and their shadow rays: the packet kernels) carry the frame.
    # They are isolated by rendering with max_depth 0 -- level 0 launches exactly the same two kernels on exactly the
    # same rays whatever the depth limit -- so each hipEvent bracket below is ONE launch of ONE kernel.
    timed_all = [step(flags=rt.RT_FLAG_TIME_KERNELS, max_depth=0) for _ in range(5)]
    counted = step(flags=rt.RT_FLAG_COUNTERS, max_depth=0)
    full = step(flags=rt.RT_FLAG_TIME_KERNELS)
    barrier()
    roofline = None
    if rank == 0:
        near_ms = float(np.mean([t["nearest_ms"] for t in timed_all[1:]]))
        shad_ms = float(np.mean([t["shadow_ms"] for t in timed_all[1:]]))
        n_sph = flat.info.n_spheres
        # algorithmic bytes: nodes read, primitive records tested, queue records read and written
        near_bytes = (counted["nodes_nearest"] * NODE_BYTES + counted["prims_nearest"] * TRI_BYTES +
                      counted["rays_nearest"] * (RAY_QUEUE_BYTES + HIT_QUEUE_BYTES))
        shad_bytes = (counted["nodes_shadow"] * NODE_BYTES + counted["prims_shadow"] * TRI_BYTES + counted["rays_shadow"] * SURF_BYTES)
        if shad_ms >= near_ms:
            # two point-type lights: the two-rays-per-lane form of the shadow packet kernel
            kname, ms, nbytes = "k_shadow_point_packet2", shad_ms, shad_bytes
        else:
            kname, ms, nbytes = "k_intersect_packet", near_ms, near_bytes
        achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # HBM-side bytes per launch of that kernel from the PMC passes (tools/pmc_passes.sh -> profiles/r01_traffic.json:
        # FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes); counters cannot be read from inside this process
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                traffic = json.load(f).get(kname, {}).get("traffic_bytes_per_launch")
        except (OSError, ValueError):
            pass
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "launch_ms": round(ms, 4), "launches_per_step": 1,
                    "algorithmic_bytes_per_launch": int(nbytes),
                    "nearest_ms": round(near_ms, 4), "shadow_ms": round(shad_ms, 4),
                    "all_levels_nearest_ms": round(full["nearest_ms"], 4), "all_levels_shadow_ms": round(full["shadow_ms"], 4),
                    "levels": int(full["levels"]),
                    "nodes_per_ray_nearest": round(counted["nodes_nearest"] / max(1, counted["rays_nearest"]), 2),
                    "nodes_per_ray_shadow": round(counted["nodes_shadow"] / max(1, counted["rays_shadow"]), 2),
                    "prims_per_ray_nearest": round(counted["prims_nearest"] / max(1, counted["rays_nearest"]), 2),
                    "prims_per_ray_shadow": round(counted["prims_shadow"] / max(1, counted["rays_shadow"]), 2),
                    "spheres": int(n_sph)}

    # ---- CPU baseline: the oracle restatement (bit-identical to the compiled reference on the bundled scenes) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle
        cw, ch = W // 4, H // 4
        ccam = rt.Camera(flat.fov, cw, ch, eye=eye, rot_x=rot_x)
        threads = min(16, os.cpu_count() or 1)
        t1 = time.perf_counter()
        _, n1, s1 = oracle.render(flat, ccam, cw, ch, seed=1, threads=1)
        d1 = time.perf_counter() - t1
        reps, tot, t2 = 0, 0, time.perf_counter()
        while time.perf_counter() - t2 < 10.0:
            _, nn, ns = oracle.render(flat, ccam, cw, ch, seed=1, threads=threads)
            tot += nn + ns
            reps += 1
        dN = time.perf_counter() - t2
        cpu = {"value": round(tot / dN / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
               "sample": f"same C2 scene and camera at {cw}x{ch} (1/16 of the pixels), full ray tree, {reps} frames in {dN:.1f} s on {threads} threads",
               "single_thread_value": round((n1 + s1) / d1 / 1e6, 3)}

    if rank == 0:
        if args.save:
            rt.write_bmp(args.save, frames[(args.steps - 1) % D].cpu().numpy())
        value = rays / elapsed / 1e6
        out = {
            "metric": "Mrays/sec (primary+secondary) at 1080p", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "frames_in_flight": D, "single_frame_latency_ms": round(latency_ms, 4), "assembled_equals_single_rank": assembled_ok,
            "config": {"workload": f"C2: {flat.info.n_tris}-triangle textured height field + 2 bump spheres, ambient+directional+point light, "
                                   f"{W}x{H}, {ax}x{ay} samples/pixel, tiles sharded over {world} GPU(s)",
                       "width": W, "height": H, "samples_per_pixel": ax * ay, "rays_per_step": rays // args.steps,
                       "eye": list(eye), "rot_x": rot_x, "max_depth": rt.MAX_REFLECTIONS},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
