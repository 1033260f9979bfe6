#!/usr/bin/env python3
"""Every BASELINE.json configuration on one MI355X with the CPU restatement timed beside it on the same box (bench.py --all-configs):

    C1  the reference's 13 bundled scenes a..m at its own 640x480, 1 sample per pixel
    C2  the bench workload (100 352-triangle textured height field + bump spheres, 1920x1080)
    C3  64 nested CSG objects, 1920x1080, depth limit 5
    C4  area lights with 64 samples each, 1920x1080
    C5  4 999 122 triangles, 3840x2160, 4x4 samples per pixel (one GPU's frame; bench.py --gpus 8 strong-scales it)

Per configuration: GPU ms per frame (frames in flight, and one frame alone), rays per frame, Mrays/s; the oracle restatement
(bit-identical to the compiled reference on the bundled scenes) on one thread and on all threads, on a BOUNDED sample of the same
workload -- the same scene and camera at a reduced resolution where the full frame would take more than a few seconds (it is a
rate: rays per second of the same ray tree) -- and the ratio.  Writes profiles/r03_configs.json; every configuration whose kernels have committed counters of this very build
(profiles/r03_config_roofline.json, tools/config_roofline.py) carries a `roofline` block for the kernels that dominate its frame.
Usage: python tools/bench_configs.py [out.json]      (CONFIGS=C1,C3 restricts; C5_GRID=500 shrinks C5 for a rehearsal)
"""
import gzip
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gpu_time(rt, torch, flat, W, H, aa, max_depth, eye, rot_x, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc = rt.Scene(flat, 0)
    upload_s = time.perf_counter() - t0                   # rt_scene_upload: sub-trees, treelets, quads, copies
    cam = rt.Camera(flat.fov, W, H, eye=eye, rot_x=rot_x)
    bufs = [torch.zeros(H, W, 3, device="cuda") for _ in range(3)]
    st = sc.render_into(bufs[0].data_ptr(), W, H, aa=aa, camera=cam, max_depth=max_depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        st = sc.render_into(bufs[0].data_ptr(), W, H, aa=aa, camera=cam, max_depth=max_depth)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / reps
    def piped_ms(flags, D):
        while len(bufs) < D:
            bufs.append(torch.zeros(H, W, 3, device="cuda"))
        pending = []
        n = D * reps
        for rep in range(2):            # (the first pass warms the mode up: the lanes' own accumulators)
            t0 = time.perf_counter()
            for i in range(n):
                if len(pending) == D:
                    sc.render_wait(pending.pop(0))
                pending.append(sc.render_submit(bufs[i % D].data_ptr(), W, H, aa=aa, camera=cam, max_depth=max_depth, flags=flags))
            while pending:
                sc.render_wait(pending.pop(0))
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    sharded = piped_ms(0, 3)                              # every frame cut into a shard per lane, three frames in flight (round 2's form)
    whole = piped_ms(rt.RT_FLAG_WHOLE_FRAME, 6)           # whole frames dealt to the lanes, six in flight
    sc.close()
    return alone, min(sharded, whole), st, sharded, whole, upload_s


def run_all(out_path, log=print):
    import torch
    import raytracer_amd as rt
    from raytracer_amd import scenes
    import oracle
    threads = min(16, os.cpu_count() or 1)
    only = [c for c in os.environ.get("CONFIGS", "").split(",") if c]
    c5_grid = int(os.environ.get("C5_GRID", "1581"))
    rows = []
    # roofline blocks per configuration: counters of the kernels alone, only if they were taken with these very sources
    from raytracer_amd import buildinfo
    roof, roof_path = {}, os.environ.get("RT_CONFIG_ROOFLINE") or os.path.join(ROOT, "profiles", "r03_config_roofline.json")
    try:
        rdoc = json.load(open(roof_path))
        if rdoc.get("source_sha") == buildinfo.source_sha():
            roof = rdoc.get("configs", {})
        else:
            log(f"{roof_path} is for sources {rdoc.get('source_sha')}, this build is {buildinfo.source_sha()}: no roofline blocks")
    except (OSError, ValueError):
        log(f"no {roof_path}: no roofline blocks")

    def roofline_of(name):
        b = roof.get(name.split("-")[0])
        if not b or name.startswith("C1"):
            return None
        keep = ("kernel", "share_of_kernel_time", "avg_launch_us", "valu_insts_per_launch", "frac_of_valu_issue", "salu_plus_branch_per_valu", "wave_wait_frac",
                "hbm_bytes_per_launch", "hbm_frac", "l2_hit_rate", "mean_resident_waves_per_simd", "launches_in_trace")
        ks = [{k: e.get(k) for k in keep} for e in b["kernels"] if e["share_of_kernel_time"] >= 0.05]
        return {"bound": "valu_issue", "peak": "1228.8 G wave-instr/s (1024 SIMDs x 2.4 GHz / 2); HBM 8000 GB/s", "kernels_alone": True,
                "dominant_kernel": b["dominant_kernel"], "frac": ks[0]["frac_of_valu_issue"] if ks else None, "kernels": ks,
                "counters_from": os.path.relpath(os.path.abspath(roof_path), ROOT), "source_sha": buildinfo.source_sha()}

    def one(name, label, flat, W, H, aa=1, max_depth=16, eye=(0.0, 0.0, 0.0), rot_x=0.0, reps=5, cpu_scale=1, cpu_aa=None, cpu_threads_only=False, load_s=None):
        alone, piped, st, sharded, whole, upload_s = gpu_time(rt, torch, flat, W, H, (aa, aa), max_depth, eye, rot_x, reps)
        rays = st["rays_nearest"] + st["rays_shadow"]
        # CPU sample: same scene, camera and depth limit; resolution divided by cpu_scale per axis (and cpu_aa samples per pixel)
        cw, ch, ca = W // cpu_scale, H // cpu_scale, (aa if cpu_aa is None else cpu_aa)
        ccam = rt.Camera(flat.fov, cw, ch, eye=eye, rot_x=rot_x)
        # (a bundled scene's frame is ~10 ms on 16 threads: frames are repeated, after a first one that warms the thread pool up, until
        # 0.3 s have gone by, and the rate is all their rays over all their time)
        t0 = time.perf_counter()
        _, nn, ns = oracle.render(flat, ccam, cw, ch, aa=ca, max_depth=max_depth, seed=1, threads=threads)
        dN = time.perf_counter() - t0
        cpu_frames = 1
        if dN < 0.3:
            cpu_frames, t0 = 0, time.perf_counter()
            while cpu_frames < 3 or time.perf_counter() - t0 < 0.3:
                oracle.render(flat, ccam, cw, ch, aa=ca, max_depth=max_depth, seed=1, threads=threads)
                cpu_frames += 1
            dN = (time.perf_counter() - t0) / cpu_frames
        cpuN = (nn + ns) / dN / 1e6
        cpu1 = None
        if not cpu_threads_only:
            # one thread on a further reduced sample when the all-thread run took long (bounded: the single-thread run is ~threads x slower)
            s1 = 1
            while dN * threads / (s1 * s1) > 20.0 and min(cw, ch) // (2 * s1) >= 16:
                s1 *= 2
            c1w, c1h = cw // s1, ch // s1
            c1cam = rt.Camera(flat.fov, c1w, c1h, eye=eye, rot_x=rot_x)
            t0 = time.perf_counter()
            _, n1, s1r = oracle.render(flat, c1cam, c1w, c1h, aa=ca, max_depth=max_depth, seed=1, threads=1)
            d1 = time.perf_counter() - t0
            cpu1 = (n1 + s1r) / d1 / 1e6
        row = {"config": name, "workload": label, "width": W, "height": H, "samples_per_pixel": aa * aa, "max_depth": max_depth,
               "rays_per_frame": int(rays), "levels": int(st["levels"]), "batch_splits": int(st["batch_splits"]),
               "load_s": None if load_s is None else round(load_s, 3), "upload_s": round(upload_s, 3),
               "gpu_ms_per_frame": round(piped * 1e3, 4), "gpu_single_frame_ms": round(alone * 1e3, 4),
               "gpu_ms_per_frame_sharded_3_in_flight": round(sharded * 1e3, 4), "gpu_ms_per_frame_whole_frames_6_in_flight": round(whole * 1e3, 4),
               "gpu_mrays_s": round(rays / piped / 1e6, 1), "gpu_single_frame_mrays_s": round(rays / alone / 1e6, 1),
               "cpu_mrays_s_1_thread": None if cpu1 is None else round(cpu1, 3), "cpu_mrays_s_all_threads": round(cpuN, 3), "cpu_threads": threads,
               "cpu_sample": f"{cw}x{ch}, {ca * ca} sample(s)/pixel, same scene, camera and depth limit ({nn + ns} rays in {dN * 1e3:.1f} ms on {threads} threads, mean of {cpu_frames} frame(s))",
               "gpu_over_cpu_all_threads": round(rays / piped / 1e6 / cpuN, 1),
               "gpu_over_cpu_1_thread": None if cpu1 is None else round(rays / piped / 1e6 / cpu1, 1),
               "roofline": roofline_of(name)}
        rows.append(row)
        log(f"{name}: GPU {row['gpu_ms_per_frame']} ms/frame (shards {row['gpu_ms_per_frame_sharded_3_in_flight']}, whole frames {row['gpu_ms_per_frame_whole_frames_6_in_flight']}, {row['gpu_single_frame_ms']} alone), {rays / 1e6:.2f} Mrays, {row['gpu_mrays_s']} Mrays/s; "
            f"CPU {row['cpu_mrays_s_1_thread']} / {row['cpu_mrays_s_all_threads']} Mrays/s (1 / {threads} threads); x{row['gpu_over_cpu_all_threads']}; set-up: load {row['load_s']} s, upload {row['upload_s']} s")

    with tempfile.TemporaryDirectory() as tmp:
        if not only or "C1" in only:
            gold = os.path.join(ROOT, "tests", "golden", "scenes")
            for ch_ in "abcdefghijklm":
                dst = os.path.join(tmp, f"{ch_}.rtscene")
                with gzip.open(os.path.join(gold, f"{ch_}.rtscene.gz"), "rb") as a, open(dst, "wb") as b:
                    shutil.copyfileobj(a, b)
                flat = rt.FlatScene.load(dst)
                one(f"C1-{ch_}", f"bundled scene {ch_}.json, 640x480, 1 sample/pixel", flat, 640, 480, reps=10, cpu_scale=4 if ch_ == "m" else 1)
        if not only or "C2" in only:
            path = scenes.c2_scene(tmp, grid=224)
            t0 = time.perf_counter(); flat = rt.FlatScene.from_json(path); load_s = time.perf_counter() - t0   # (scene file -> flat scene: parsing, the reference's tree)
            one("C2", "100 352-triangle textured height field + 2 bump spheres, 1920x1080", flat, 1920, 1080, eye=(0.0, 1.5, 1.0), rot_x=-0.3, reps=10, cpu_scale=4, load_s=load_s)
        if not only or "C3" in only:
            path = scenes.c3_scene(tmp)
            t0 = time.perf_counter(); flat = rt.FlatScene.from_json(path); load_s = time.perf_counter() - t0   # (scene file -> flat scene: parsing, the reference's tree)
            one("C3", "64 nested CSG objects (half reflective, a quarter refractive), 1920x1080, depth limit 5", flat, 1920, 1080, max_depth=5,
                eye=(0.0, 1.0, 2.0), rot_x=-0.25, reps=5, cpu_scale=4, load_s=load_s)
        if not only or "C4" in only:
            path = scenes.c4_scene(tmp, 64)
            t0 = time.perf_counter(); flat = rt.FlatScene.from_json(path); load_s = time.perf_counter() - t0   # (scene file -> flat scene: parsing, the reference's tree)
            one("C4", "two area lights x 64 shadow samples per hit, 1920x1080", flat, 1920, 1080, reps=3, cpu_scale=8, load_s=load_s)
        if not only or "C5" in only:
            path = scenes.c5_scene(tmp, c5_grid)
            t0 = time.perf_counter(); flat = rt.FlatScene.from_json(path); load_s = time.perf_counter() - t0   # (scene file -> flat scene: parsing, the reference's tree)
            one("C5", f"{flat.info.n_tris}-triangle height field, 3840x2160, 4x4 samples/pixel", flat, 3840, 2160, aa=4, eye=(0.0, 1.5, 1.0), rot_x=-0.3,
                reps=1, cpu_scale=8, cpu_aa=1, load_s=load_s)
    doc = {"note": "GPU: one MI355X, frames in flight (rt_render_submit: the better of shards with 3 outstanding and whole frames, RT_FLAG_WHOLE_FRAME, with 6; both given) and one frame alone (rt_render); rays = findNearest + calcShadow calls. "
                   "CPU: oracle/restate.cpp (the reference's algorithm, one ray at a time) on this box's host cores.",
           "cpu_threads": threads, "configs": rows}
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    return out_path


if __name__ == "__main__":
    run_all(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_configs.json"))
