#!/usr/bin/env python3
"""bench.py -- Mrays/s of the MI355X trace path on BASELINE.json's 1080p workload (config 2).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one frame of the C2 scene (100 352-triangle textured height field + two bump-mapped spheres, ambient +
directional + point light) at 1920x1080: generate -> intersect -> shade -> shadow -> resolve on the GPU, scene and
queues resident in HBM, nothing crossing PCIe inside the timed region.  With N > 1 ranks the frame is sharded by
interleaved 32x8 tiles (scene replicated per GPU), each rank packs its tiles and rank 0 gathers them over RCCL and
unpacks them into the full frame -- that exchange is inside the timed region.  `value` scales WEAKLY: the sample grid
grows with N (1x1, 2x1, 2x2, 4x2 samples per pixel for N = 1, 2, 4, 8) so every GPU traces one sample per pixel of
its share, i.e. 1920*1080 primary rays per GPU at every N.

With N > 1 the same run also times the multi-GPU configurations BASELINE.json names, STRONG-scaled (one frame of fixed size
cut over the ranks, gather in the timed region), and reports them in `baseline_configs`: C4 (area lights, 64 shadow samples
per hit, 1080p) for N = 2 .. 7 and C5 (4 999 122 triangles, 3840x2160, 4x4 samples per pixel) for N >= 8.
`--all-configs` (N = 1) times every BASELINE configuration with a CPU baseline beside it (tools/bench_configs.py).

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
# Vector-instruction issue ceiling (MI355X_MICROARCH.md, cycle constants: a wave64 VALU instruction occupies its SIMD for 2 cycles):
# 256 CUs x 4 SIMDs x 2.4 GHz / 2.  tools/ubench/valu_issue.hip measures 1.10e12 for v_mul_f32 and half that for packed f32 on the box.
SIMDS, CLOCK_GHZ, VALU_ISSUE_CYCLES = 1024, 2.4, 2.0
VALU_PEAK_GIPS = SIMDS * CLOCK_GHZ / VALU_ISSUE_CYCLES
PMC_FILE = os.environ.get("RT_PMC_FILE") or os.path.join(ROOT, "profiles", "r03_pmc_counters.json")   # (the override: tools/evidence_r03.sh, same box, same build)
# algorithmic bytes per unit of traversal work (with this build's 64 B two-child node)
NODE_BYTES, TRI_BYTES, RAY_QUEUE_BYTES, HIT_QUEUE_BYTES, SURF_BYTES = 64, 64, 44, 12, 84


class Workload:
    def __init__(self, name, label, flat, W, H, aa, max_depth, eye=(0.0, 0.0, 0.0), rot_x=0.0):
        self.name, self.label, self.flat, self.W, self.H, self.aa, self.max_depth, self.eye, self.rot_x = name, label, flat, W, H, aa, max_depth, eye, rot_x


def make_workload(kind, args, world, rt, scenes, tmp):
    if kind == "c2":
        flat = rt.FlatScene.from_json(scenes.c2_scene(tmp, grid=args.grid))
        ax, ay = AA_GRID.get(world, (world, 1))
        return Workload("C2", f"C2: {flat.info.n_tris}-triangle textured height field + 2 bump spheres, ambient+directional+point light, "
                        f"{args.width}x{args.height}, {ax}x{ay} samples/pixel, tiles sharded over {world} GPU(s)",
                        flat, args.width, args.height, (ax, ay), rt.MAX_REFLECTIONS, (0.0, 1.5, 1.0), -0.3)
    if kind == "c4":
        flat = rt.FlatScene.from_json(scenes.c4_scene(tmp, args.c4_samples))
        return Workload("C4", f"C4: scene m's geometry with two area lights of {args.c4_samples} samples each, {args.width}x{args.height}, 1 sample/pixel, "
                        f"one frame strong-scaled over {world} GPU(s)", flat, args.width, args.height, (1, 1), rt.MAX_REFLECTIONS)
    if kind == "c5":
        flat = rt.FlatScene.from_json(scenes.c5_scene(tmp, args.c5_grid))
        W, H = 2 * args.width, 2 * args.height
        return Workload("C5", f"C5: {flat.info.n_tris}-triangle height field, directional + point light, {W}x{H}, 4x4 samples/pixel, "
                        f"one frame strong-scaled over {world} GPU(s)", flat, W, H, (4, 4), rt.MAX_REFLECTIONS, (0.0, 1.5, 1.0), -0.3)
    raise ValueError(kind)


class Runner:
    """One workload on this rank's GPU: frames in flight, and with several ranks the tile exchange to rank 0."""

    def __init__(self, ctx, wl, D, rgb8=False):
        self.ctx, self.wl, self.rgb8 = ctx, wl, rgb8
        rt, torch = ctx.rt, ctx.torch
        self.scene = rt.Scene(wl.flat, ctx.local_rank)
        self.cam = rt.Camera(wl.flat.fov, wl.W, wl.H, eye=wl.eye, rot_x=wl.rot_x)
        self.D = D
        dev, W, H, world, rank = ctx.dev, wl.W, wl.H, ctx.world, ctx.rank
        self.frames = [torch.zeros(H, W, 3, device=dev, dtype=torch.float32) for _ in range(D)]
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        # gather payload: every rank's buffer padded to the largest tile count (8100 tiles do not divide by 8), so that the
        # exchange is ONE grouped gather -- all peers send to the root at once, each over its own xGMI link -- instead of a
        # chain of point-to-point receives.  float RGB (12 B per pixel) or, with --rgb8, bytes quantised on the device (3 B).
        self.pad = max(self.scene.tiles_count(W, H, r, world) for r in range(world)) * 256 * 3
        pdt = torch.uint8 if rgb8 else torch.float32
        self.packed = torch.zeros(self.pad, device=dev, dtype=pdt)
        self.sizes = [self.pad] * world
        self.recv = [torch.zeros(self.pad, device=dev, dtype=pdt) for _ in range(world)] if (world > 1 and rank == 0) else None
        self.frame8 = torch.zeros(H, W, 3, device=dev, dtype=torch.uint8) if (rgb8 and rank == 0) else None
        # The exchange runs on a stream of its own.  Lane 0 of the renderer works on the caller's stream (torch's current one):
        # a gather queued there would hold that lane's next shards back until every rank had arrived.  Order is kept with two
        # events per frame buffer: the side stream starts after the frame is complete (rt_render_wait has waited for it on the
        # host), and the next frame rendered into the same buffer starts after that buffer's tiles were packed and unpacked.
        self.side = torch.cuda.Stream(device=dev) if world > 1 else None
        self.assembled = [None] * D
        # With several frames in flight each frame goes to ONE of the handle's lanes (RT_FLAG_WHOLE_FRAME) instead of being cut into a shard
        # per lane: a launch over a whole frame amortises the tail of its longest walks over four times the rays.  --shards: the round-2 form.
        self.submit_flags = ctx.rt.RT_FLAG_WHOLE_FRAME if (D > 1 and not getattr(ctx, "shards", False)) else 0

    def close(self):
        self.scene.close()

    def assemble(self, slot):
        # tiles -> rank 0 (one grouped RCCL gather over xGMI, every peer sending to the root at once), then de-interleave
        ctx, wl, sc = self.ctx, self.wl, self.scene
        torch, W, H, rank, world = ctx.torch, wl.W, wl.H, ctx.rank, ctx.world
        buf = self.frames[slot]
        with torch.cuda.stream(self.side):
            ss = self.side.cuda_stream
            if self.rgb8:
                sc.tiles_pack_rgb8(buf.data_ptr(), self.packed.data_ptr(), W, H, rank, world, stream=ss)
            else:
                sc.tiles_pack(buf.data_ptr(), self.packed.data_ptr(), W, H, rank, world, stream=ss)
            if ctx.rehearsal:
                host = ctx.rd.gather_packed(self.packed.cpu(), self.sizes, rank, world)
                bufs = [h.to(ctx.dev) for h in host] if rank == 0 else None
            else:
                bufs = ctx.rd.gather_packed(self.packed, self.sizes, rank, world, bufs=self.recv)
            if rank == 0:
                if self.rgb8:
                    sc.tiles_quantize_rgb8(buf.data_ptr(), self.frame8.data_ptr(), W, H, 0, world, stream=ss)
                for r in range(1, world):
                    if self.rgb8:
                        sc.tiles_unpack_rgb8(bufs[r].data_ptr(), self.frame8.data_ptr(), W, H, r, world, stream=ss)
                    else:
                        sc.tiles_unpack(bufs[r].data_ptr(), buf.data_ptr(), W, H, r, world, stream=ss)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.assembled[slot] = ev

    def buffer_free(self, slot):
        """The caller's stream (and with it the frame about to be submitted) waits until the buffer's last exchange is through."""
        if self.assembled[slot] is not None:
            self.ctx.torch.cuda.current_stream(self.ctx.dev).wait_event(self.assembled[slot])
            self.assembled[slot] = None

    def step(self, flags=0, max_depth=None):
        """One synchronous frame."""
        wl, ctx = self.wl, self.ctx
        self.buffer_free(0)
        st = self.scene.render_into(self.frames[0].data_ptr(), wl.W, wl.H, aa=wl.aa, camera=self.cam, tile_rank=ctx.rank, tile_ranks=ctx.world, stream=self.stream,
                                    flags=flags, max_depth=wl.max_depth if max_depth is None else max_depth)
        if ctx.world > 1:
            self.assemble(0)
            self.side.synchronize()
        return st

    def run_steps(self, k):
        """k frames with up to D in flight: frame i+1 is queued before frame i is waited for (and, with N > 1 ranks, gathered)."""
        wl, ctx, sc = self.wl, self.ctx, self.scene
        rays, pending = 0, []
        self.state_cases = 0

        def retire():
            nonlocal rays
            ticket, slot = pending.pop(0)
            st = sc.render_wait(ticket)
            if ctx.world > 1:
                self.assemble(slot)
            rays += st["rays_nearest"] + st["rays_shadow"]
            self.state_cases = st["state_cases"]            # of the last frame (every frame is the same frame)
        for i in range(k):
            if len(pending) == self.D:
                retire()
            slot = i % self.D
            self.buffer_free(slot)
            pending.append((sc.render_submit(self.frames[slot].data_ptr(), wl.W, wl.H, aa=wl.aa, camera=self.cam, tile_rank=ctx.rank, tile_ranks=ctx.world,
                                             stream=self.stream, max_depth=wl.max_depth, flags=self.submit_flags), slot))
        while pending:
            retire()
        return rays

    def timed(self, steps, warmup, min_seconds=0.5, max_blocks=200):
        """Timed blocks of exactly `steps` frames each, every block bracketed by a barrier + device synchronisation on both sides and
        taken as the MAX over ranks; blocks are repeated until `min_seconds` of timed work have gone by (a 20-frame block of C2 is 13 ms:
        too short for the SMI sampler and exposed to the pipelined rate's run-to-run modes).  Returns (seconds of the MEDIAN block, rays
        of the whole job in one block, latency of a frame rendered alone in ms, every block's seconds)."""
        ctx = self.ctx
        self.run_steps(warmup)
        blocks, rays, total = [], 0, 0.0
        while True:
            ctx.barrier()
            t0 = time.perf_counter()
            rays = self.run_steps(steps)
            ctx.barrier()
            elapsed = ctx.max_over_ranks(time.perf_counter() - t0)      # the same number on every rank: all take the same decision below
            blocks.append(elapsed)
            total += elapsed
            if total >= min_seconds or len(blocks) >= max_blocks:
                break
        ctx.barrier()
        t1 = time.perf_counter()
        reps = 5 if blocks[0] / max(steps, 1) < 0.02 else 2
        for _ in range(reps):
            self.step()
        ctx.barrier()
        latency_ms = (time.perf_counter() - t1) / reps * 1e3
        return float(np.median(blocks)), ctx.sum_over_ranks(rays), latency_ms, blocks

    def assembled_equals_single_rank(self):
        """Outside the timed region: the frame assembled from every rank's tiles must be, bit for bit, the frame one rank renders alone
        (with --rgb8: its 8-bit form as rt_quantize_rgb8 makes it)."""
        ctx, wl, torch = self.ctx, self.wl, self.ctx.torch
        if ctx.world == 1:
            return None
        ok = None
        self.step()
        if ctx.rank == 0:
            alone = torch.zeros(wl.H, wl.W, 3, device=ctx.dev, dtype=torch.float32)
            self.scene.render_into(alone.data_ptr(), wl.W, wl.H, aa=wl.aa, camera=self.cam, tile_rank=0, tile_ranks=1, stream=self.stream, max_depth=wl.max_depth)
            if self.rgb8:
                want = torch.from_numpy(ctx.rt.quantize_rgb8(alone.cpu().numpy()))
                ok = bool(torch.equal(want, self.frame8.cpu()))
            else:
                ok = bool(torch.equal(alone.view(torch.int32), self.frames[0].view(torch.int32)))
            del alone
        ctx.barrier()
        return ok


class Ctx:
    pass


def one_gpu_ms(ctx, wl, D, steps):
    """The same workload -- same frame size, same samples per pixel -- on rank 0's GPU alone, D frames in flight, no exchange: ms per frame
    (None on the other ranks, which wait at the barrier).  Outside every timed region."""
    torch = ctx.torch
    ms = None
    if ctx.rank == 0:
        world_saved, rank_saved = ctx.world, ctx.rank
        ctx.world = 1
        try:
            solo = Runner(ctx, wl, D)
            solo.run_steps(2)
            torch.cuda.synchronize(ctx.dev)
            t0 = time.perf_counter()
            solo.run_steps(steps)
            torch.cuda.synchronize(ctx.dev)
            ms = (time.perf_counter() - t0) / steps * 1e3
            solo.close()
        finally:
            ctx.world, ctx.rank = world_saved, rank_saved
    ctx.barrier()
    return ms


def load_pmc():
    """Counters of the final build's kernels (tools/evidence_r03.sh -> tools/config_roofline.py): they cannot be read from inside this process,
    so they come from the committed file -- and only if it was taken with these very sources."""
    from raytracer_amd import buildinfo
    try:
        with open(PMC_FILE) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return None, "no counter file"
    if pmc.get("source_sha") != buildinfo.source_sha():
        return None, f"counter file is for sources {pmc.get('source_sha')}, this build is {buildinfo.source_sha()}"
    return pmc, None


def roofline_block(runner, flat):
    """Roofline of the dominant kernel.  The two traversal kernels of level 0 (primary rays and their shadow rays: the packet
    kernels) carry the frame.  They are isolated by rendering with max_depth 0 -- level 0 launches exactly the same two kernels
    on exactly the same rays whatever the depth limit -- so each hipEvent bracket is ONE launch of ONE kernel, timed live on the
    stream it runs on.  What binds them is vector-instruction issue, not memory (the scene is cache-resident and a packet
    fetches a node once for 64 rays): achieved = VALU wave-instructions per launch (SQ_INSTS_VALU of the committed counter pass
    of this build) / the live launch time, peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction."""
    import raytracer_amd as rt
    timed_all = [runner.step(flags=rt.RT_FLAG_TIME_KERNELS, max_depth=0) for _ in range(5)]
    counted = runner.step(flags=rt.RT_FLAG_COUNTERS, max_depth=0)
    full = runner.step(flags=rt.RT_FLAG_TIME_KERNELS)
    runner.ctx.barrier()
    if runner.ctx.rank != 0:
        return None
    near_ms = float(np.mean([t["nearest_ms"] for t in timed_all[1:]]))
    shad_ms = float(np.mean([t["shadow_ms"] for t in timed_all[1:]]))
    near_bytes = (counted["nodes_nearest"] * NODE_BYTES + counted["prims_nearest"] * TRI_BYTES + counted["rays_nearest"] * (RAY_QUEUE_BYTES + HIT_QUEUE_BYTES))
    shad_bytes = (counted["nodes_shadow"] * NODE_BYTES + counted["prims_shadow"] * TRI_BYTES + counted["rays_shadow"] * SURF_BYTES)
    pmc, why = load_pmc()
    kernels = (pmc or {}).get("kernels", {})
    # the kernel names of this build's level-0 launches come with the counter file (the fused and the queued forms differ)
    near_name = (pmc or {}).get("nearest_kernel", "k_intersect_packet")
    shad_name = (pmc or {}).get("shadow_kernel", "k_shadow_point_packet2")
    if shad_ms >= near_ms:
        kname, ms, alg_bytes, rays = shad_name, shad_ms, shad_bytes, counted["rays_shadow"]
        busy = counted["nodes_shadow"] / max(1, 64 * counted["steps_shadow"])
    else:
        kname, ms, alg_bytes, rays = near_name, near_ms, near_bytes, counted["rays_nearest"]
        busy = counted["nodes_nearest"] / max(1, 64 * counted["steps_nearest"])
    k = kernels.get(kname)
    out = {"bound": "valu_issue", "kernel": kname, "achieved": None, "peak": round(VALU_PEAK_GIPS, 1), "unit": "G wave-instr/s", "frac": None,
           "traffic": None, "hbm_frac": None, "launch_ms": round(ms, 4), "launches_per_step": 1,
           "peak_is": f"{SIMDS} SIMDs x {CLOCK_GHZ} GHz / {VALU_ISSUE_CYCLES:g} cycles per wave64 VALU instruction"}
    if k:
        valu = k["SQ_INSTS_VALU"]
        achieved = valu / (ms * 1e-3) / 1e9
        traffic = k.get("traffic_bytes_per_launch")
        wc = k.get("SQ_WAVE_CYCLES")
        out.update({"achieved": round(achieved, 1), "frac": round(achieved / VALU_PEAK_GIPS, 4),
                    "valu_insts_per_launch": int(valu), "valu_insts_per_ray": round(valu / max(1, rays), 2),
                    "salu_insts_per_launch": int(k.get("SQ_INSTS_SALU", 0)),
                    # share of a wave's resident time parked in s_waitcnt / at barriers, stalled at issue, and issuing (quad-cycle counters, disjoint)
                    "wave_wait_frac": round(k["SQ_WAIT_ANY"] / wc, 3) if wc else None,
                    "wave_issue_stall_frac": round(k["SQ_WAIT_INST_ANY"] / wc, 3) if wc else None,
                    "wave_active_frac": round(k["SQ_ACTIVE_INST_ANY"] / wc, 3) if wc else None,
                    # the profiler's own VALUBusy: quad-cycles in which a SIMD had a vector instruction in flight (SQ_ACTIVE_INST_VALU x 4 cycles)
                    # over the SIMD-cycles of the launch -- the counter books about one quad-cycle per wave64 instruction, i.e. twice the 2 cycles
                    # `frac` prices an instruction at, so this reads about twice `frac`; the truth for this instruction mix lies between the two
                    "valu_busy_frac": round(k["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * ms * 1e-3 * CLOCK_GHZ * 1e9), 4) if k.get("SQ_ACTIVE_INST_VALU") else None,
                    "counter_launch_ms": round(k["GRBM_GUI_ACTIVE"] / 8 / (CLOCK_GHZ * 1e6), 4) if k.get("GRBM_GUI_ACTIVE") else None,
                    "traffic": traffic,
                    "hbm_achieved_gbs": round(traffic / (ms * 1e-3) / 1e9, 1) if traffic else None,
                    "hbm_frac": round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                    "l2_hit_rate": round(k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]), 4) if k.get("TCC_HIT_sum") is not None and (k.get("TCC_HIT_sum", 0) + k.get("TCC_MISS_sum", 0)) > 0 else None,
                    # mean waves resident per SIMD over the launch (SQ_WAVE_CYCLES is in quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs)
                    "mean_resident_waves_per_simd": round(wc * 4 / (k["GRBM_GUI_ACTIVE"] / 8) / SIMDS, 2) if wc and k.get("GRBM_GUI_ACTIVE") else None,
                    "counters_from": os.path.relpath(os.path.abspath(PMC_FILE), ROOT)})
        # the whole step against the same two ceilings: every kernel's instructions and HBM bytes over the pipelined frame time
        out["_step_valu"] = sum(v.get("SQ_INSTS_VALU", 0) * v.get("launches_per_frame", 1) for v in kernels.values())
        out["_step_valu_busy"] = sum(v.get("SQ_ACTIVE_INST_VALU", 0) * v.get("launches_per_frame", 1) for v in kernels.values())
        out["_step_traffic"] = sum((v.get("traffic_bytes_per_launch") or 0) * v.get("launches_per_frame", 1) for v in kernels.values())
    else:
        out["counters_missing"] = why or f"{kname} is not in the counter file"
    out.update({"busy_lane_frac": round(busy, 3),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "algorithmic_bytes_note": "per-RAY bytes of SURVEY 8d (64 B x nodes + 64 B x triangle tests + queue records); a packet fetches a node once for 64 rays "
                                          "from LDS, so these are not HBM bytes and no fraction is taken of them",
                "nearest_ms": round(near_ms, 4), "shadow_ms": round(shad_ms, 4),
                "all_levels_nearest_ms": round(full["nearest_ms"], 4), "all_levels_shadow_ms": round(full["shadow_ms"], 4), "levels": int(full["levels"]),
                "nodes_per_ray_nearest": round(counted["nodes_nearest"] / max(1, counted["rays_nearest"]), 2),
                "nodes_per_ray_shadow": round(counted["nodes_shadow"] / max(1, counted["rays_shadow"]), 2),
                "prims_per_ray_nearest": round(counted["prims_nearest"] / max(1, counted["rays_nearest"]), 2),
                "prims_per_ray_shadow": round(counted["prims_shadow"] / max(1, counted["rays_shadow"]), 2),
                "steps_per_packet_nearest": round(counted["steps_nearest"] / max(1, counted["rays_nearest"] / 64), 1),
                "steps_per_packet_shadow": round(counted["steps_shadow"] / max(1, counted["rays_shadow"] / 64), 1),
                "spheres": int(flat.info.n_spheres)})
    return out


def cpu_baseline_block(rt, wl, seconds=10.0):
    """The oracle restatement (bit-identical to the compiled reference on the bundled scenes) on this host's cores: same scene and
    camera at a quarter of the resolution per axis, one thread once and all threads for ~10 s."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    cw, ch = wl.W // 4, wl.H // 4
    ccam = rt.Camera(wl.flat.fov, cw, ch, eye=wl.eye, rot_x=wl.rot_x)
    threads = min(16, os.cpu_count() or 1)
    t1 = time.perf_counter()
    _, n1, s1 = oracle.render(wl.flat, ccam, cw, ch, seed=1, threads=1)
    d1 = time.perf_counter() - t1
    reps, tot, t2 = 0, 0, time.perf_counter()
    while time.perf_counter() - t2 < seconds:
        _, nn, ns = oracle.render(wl.flat, ccam, cw, ch, seed=1, threads=threads)
        tot += nn + ns
        reps += 1
    dN = time.perf_counter() - t2
    return {"value": round(tot / dN / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"same C2 scene and camera at {cw}x{ch} (1/16 of the pixels), full ray tree, {reps} frames in {dN:.1f} s on {threads} threads",
            "single_thread_value": round((n1 + s1) / d1 / 1e6, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--grid", type=int, default=224, help="height-field cells per side (224 -> 100 352 triangles)")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="repeat the timed block of --steps frames until this much timed work has gone by; the median block is reported")
    ap.add_argument("--inflight", type=int, default=0, help="frames in flight (rt_render_submit / rt_render_wait); 1 = one synchronous frame at a time; default 6 (3 with --shards)")
    ap.add_argument("--shards", action="store_true", help="cut every frame into one shard per lane even with several frames in flight (default then: whole frames dealt to the lanes)")
    ap.add_argument("--rgb8", action="store_true", help="N > 1: quantise to 8-bit pixels on the device before the gather (3 B instead of 12 B per pixel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel timing / counting passes (counter runs under rocprofv3)")
    ap.add_argument("--no-baseline-configs", action="store_true", help="N > 1: skip the strong-scaled C4 / C5 frames")
    ap.add_argument("--c4-samples", type=int, default=64)
    ap.add_argument("--c5-grid", type=int, default=1581, help="C5 height-field cells per side (1581 -> 4 999 122 triangles)")
    ap.add_argument("--config-steps", type=int, default=10, help="timed frames of the C4 / C5 runs")
    ap.add_argument("--all-configs", action="store_true", help="N = 1: also time every BASELINE configuration with a CPU baseline (tools/bench_configs.py -> profiles/r03_configs.json)")
    ap.add_argument("--save", default=None, help="write the last frame as a BMP (rank 0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd import scenes
    from raytracer_amd import distributed as rd

    ctx = Ctx()
    ctx.rt, ctx.torch, ctx.dist, ctx.rd = rt, torch, dist, rd
    ctx.shards = args.shards
    ctx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.rank = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: RT_BENCH_REHEARSAL=1 puts every rank on device 0 and gathers through the host over gloo
    ctx.rehearsal = rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    ctx.local_rank = local_rank
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
    ctx.dev = dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    cdev = "cpu" if rehearsal else dev

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(n):
        if world == 1:
            return n
        t = torch.tensor([n], device=cdev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())
    ctx.barrier, ctx.max_over_ranks, ctx.sum_over_ranks = barrier, max_over_ranks, sum_over_ranks

    # frames in flight: whole frames dealt to the lanes want a frame queued behind the one each lane is working on (default 6: C2 0.53 ms per
    # frame against 0.59 ms in shards); in shards three is what fills the lanes (more changed nothing in round 2)
    D = max(1, min(3 if args.shards else 7, args.inflight if args.inflight > 0 else (3 if args.shards else 6)))
    # ---- the metric's workload: C2, generated deterministically in the reference's JSON schema, loaded by the library's choose_scene restatement ----
    with tempfile.TemporaryDirectory() as tmp:
        wl = make_workload("c2", args, world, rt, scenes, tmp)
    runner = Runner(ctx, wl, D, rgb8=args.rgb8)
    elapsed, rays, latency_ms, blocks = runner.timed(args.steps, args.warmup, min_seconds=args.min_seconds)
    state_cases = ctx.sum_over_ranks(int(runner.state_cases))
    assembled_ok = runner.assembled_equals_single_rank()
    # N > 1: `value` scales weakly through the sample grid (N samples per pixel), and frames with >= 3 samples per pixel take the row-sum
    # path, which is faster PER RAY than the 1-sample frame of N = 1 -- value(N) / (N * value(1)) would read above
    # 100 % for that reason alone.  So the same N-sample frame is also rendered on rank 0's GPU by itself, outside the timed region, and
    # efficiency = one_gpu_same_workload_ms / (N * ms_per_step) compares like with like.
    same_ms = one_gpu_ms(ctx, wl, D, max(10, args.steps // 2)) if world > 1 else None
    roofline = None if args.no_roofline else roofline_block(runner, wl.flat)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_block(rt, wl)
    if rank == 0 and args.save:
        rt.write_bmp(args.save, runner.frames[(args.steps - 1) % D].cpu().numpy())
    runner.close()
    del runner

    # ---- N > 1: the multi-GPU configurations BASELINE.json names, one frame strong-scaled over the ranks ----
    extra = None
    if world > 1 and not args.no_baseline_configs:
        kind = "c5" if world >= 8 else "c4"
        try:
            with tempfile.TemporaryDirectory() as tmp:
                wl2 = make_workload(kind, args, world, rt, scenes, tmp)
            r2 = Runner(ctx, wl2, D, rgb8=args.rgb8)
            e2, rays2, lat2, blocks2 = r2.timed(args.config_steps, 2, min_seconds=0.0)
            ok2 = r2.assembled_equals_single_rank()
            # the same frame on rank 0 alone, for the strong-scaling ratio (the other ranks wait at the barrier)
            alone_ms = one_gpu_ms(ctx, wl2, D, max(2, args.config_steps // 2))
            r2.close()
            ms2 = e2 / args.config_steps * 1e3
            extra = {wl2.name: {"workload": wl2.label, "scaling": "strong", "n_gpus": world, "steps": args.config_steps,
                                "ms_per_frame": round(ms2, 4), "value": round(rays2 / e2 / 1e6, 2), "unit": "Mrays/s",
                                "rays_per_frame": rays2 // args.config_steps, "single_frame_latency_ms": round(lat2, 4),
                                "one_gpu_ms_per_frame": None if alone_ms is None else round(alone_ms, 4),
                                "speedup_over_one_gpu": None if alone_ms is None else round(alone_ms / ms2, 3),
                                "assembled_equals_single_rank": ok2, "gather_in_timed_region": True, "exchange": "rgb8" if args.rgb8 else "f32"}}
        except Exception as e:   # the headline line must not be lost to a failure in the extra runs
            extra = {"error": f"{type(e).__name__}: {e}"}

    configs_file = None
    if args.all_configs and world == 1 and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs
        configs_file = bench_configs.run_all(os.path.join(ROOT, "profiles", "r03_configs.json"))

    if rank == 0:
        value = rays / elapsed / 1e6
        if roofline and roofline.get("_step_valu") is not None:
            sv, stf, svb = roofline.pop("_step_valu"), roofline.pop("_step_traffic"), roofline.pop("_step_valu_busy", 0)
            ms_step = elapsed / args.steps * 1e3
            roofline["step_check"] = {"note": "all kernels of a frame over the pipelined frame time (N = 1 only: the counters are one GPU's)",
                                      "valu_insts_per_step": int(sv), "valu_frac": round(sv / (ms_step * 1e-3) / 1e9 / VALU_PEAK_GIPS, 4) if world == 1 else None,
                                      "valu_busy_frac": round(svb * 4 / (SIMDS * ms_step * 1e-3 * CLOCK_GHZ * 1e9), 4) if world == 1 and svb else None,
                                      "hbm_bytes_per_step": int(stf), "hbm_frac": round(stf / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if world == 1 else None}
        out = {
            "metric": "Mrays/sec (primary+secondary) at 1080p", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "blocks": len(blocks), "ms_per_step_min": round(min(blocks) / args.steps * 1e3, 4), "ms_per_step_max": round(max(blocks) / args.steps * 1e3, 4),
            "timing": f"median of {len(blocks)} blocks of {args.steps} steps, each bracketed by barrier + synchronize, max over ranks",
            "one_gpu_same_workload_ms": None if same_ms is None else round(same_ms, 4),
            "efficiency": None if same_ms is None else round(same_ms / (world * (elapsed / args.steps * 1e3)), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "frames_in_flight": D, "frame_to_lanes": "shards" if (D == 1 or args.shards) else "whole frames (RT_FLAG_WHOLE_FRAME)", "single_frame_latency_ms": round(latency_ms, 4), "assembled_equals_single_rank": assembled_ok,
            "exchange": None if world == 1 else ("rgb8" if args.rgb8 else "f32"),
            "config": {"workload": wl.label, "width": wl.W, "height": wl.H, "samples_per_pixel": wl.aa[0] * wl.aa[1], "rays_per_step": rays // args.steps,
                       "eye": list(wl.eye), "rot_x": wl.rot_x, "max_depth": wl.max_depth},
            # rt_stats.state_cases of one frame: exit searches of transmitted rays that found no exit, where the reference goes on from hit caches
            # earlier pixels left (objects.cpp:123-125) and the device cannot; at most this many rays of the frame can differ for that reason
            "state_cases_per_step": state_cases, "state_cases_frac_of_pixels": round(state_cases / (wl.W * wl.H), 9),
            "roofline": roofline, "cpu_baseline": cpu, "baseline_configs": extra,
        }
        if configs_file:
            out["all_configs"] = os.path.relpath(configs_file, ROOT)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
