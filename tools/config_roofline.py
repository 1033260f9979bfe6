#!/usr/bin/env python3
"""Roofline figures per BASELINE configuration from what tools/evidence_r03.sh collected (run on the GPU box, or here on the merged files):

    <root>/pmc_<CFG>/<pass>/**/*counter_collection.csv      rocprofv3 --pmc, one run per counter set (kernels alone: RT_LANES=1)
    <root>/trace_<CFG>/**/*kernel_stats.csv                 rocprofv3 --kernel-trace --stats of the same command

For every kernel that carries a configuration's frame it gives, per launch: duration (kernel trace; the counter pass's own
GRBM_GUI_ACTIVE beside it), vector instructions and their rate against the SIMDs' issue ceiling (1024 SIMDs x 2.4 GHz / 2 cycles per
wave64 instruction), scalar + branch instructions per vector instruction, the share of wave cycles spent waiting, HBM-side bytes
(FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md) against 8 TB/s, the L2 hit rate TCC_HIT / (TCC_HIT + TCC_MISS) and the mean
number of waves resident per SIMD over the launch (SQ_WAVE_CYCLES is in quad-cycles: x 4 / launch cycles / 1024 SIMDs).

Writes <root>/config_roofline.json (all configurations), <root>/pmc_counters.json (C2, the file bench.py reads its numerators from),
<root>/config_roofline.txt (the table) -- every file carries the sha of the sources the library was built from.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIMDS, CLOCK_GHZ, VALU_ISSUE_CYCLES, HBM_PEAK_GBS = 1024, 2.4, 2.0, 8000.0
VALU_PEAK_GIPS = SIMDS * CLOCK_GHZ / VALU_ISSUE_CYCLES
COMMANDS = {"C2": "python3 bench.py --steps 3 --warmup 1 --inflight 1 --min-seconds 0 --no-cpu-baseline --no-roofline   (trace: --steps 10 --warmup 2)",
            "C3": "CONFIG=C3 python3 tools/config_times.py", "C4": "CONFIG=C4 python3 tools/config_times.py",
            "C5": "CONFIG=C5 C5_GRID=1581 C5_AA=4 python3 tools/config_times.py"}


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def read_counters(d):
    """kernel (template variant) -> counter -> [value per dispatch]"""
    table = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            table[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return table


def read_trace(d):
    """kernel (template variant) -> (calls, total ns, avg ns, min ns, max ns)"""
    res = {}
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        return res
    for r in csv.DictReader(open(files[0])):
        res[short(r["Name"])] = (int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"]))
    return res


def config_block(root, cfg):
    table = read_counters(os.path.join(root, f"pmc_{cfg}"))
    trace = read_trace(os.path.join(root, f"trace_{cfg}"))
    kernels = {k: v for k, v in trace.items() if k.startswith("k_")}
    total_ns = sum(v[1] for v in kernels.values())
    rows = []
    for k, (calls, tot, avg, mn, mx) in sorted(kernels.items(), key=lambda kv: -kv[1][1]):
        c = table.get(k)
        if not c or tot / max(total_ns, 1) < 0.01:
            continue
        mean = {name: sum(v) / len(v) for name, v in c.items()}
        valu = mean.get("SQ_INSTS_VALU")
        e = {"kernel": k, "launches_in_trace": calls, "share_of_kernel_time": round(tot / total_ns, 4), "avg_launch_us": round(avg / 1e3, 2),
             "min_launch_us": round(mn / 1e3, 2), "max_launch_us": round(mx / 1e3, 2), "dispatches_in_counter_pass": max(len(v) for v in c.values())}
        if mean.get("GRBM_GUI_ACTIVE"):
            e["counter_pass_launch_us"] = round(mean["GRBM_GUI_ACTIVE"] / 8 / (CLOCK_GHZ * 1e3), 2)
        if valu is not None:
            rate = valu / avg   # instructions per ns = G/s
            e.update({"valu_insts_per_launch": int(valu), "valu_g_per_s": round(rate, 1), "frac_of_valu_issue": round(rate / VALU_PEAK_GIPS, 4)})
            if "SQ_INSTS_SALU" in mean:
                e["salu_per_valu"] = round(mean["SQ_INSTS_SALU"] / max(valu, 1), 3)
                e["salu_plus_branch_per_valu"] = round((mean["SQ_INSTS_SALU"] + mean.get("SQ_INSTS_BRANCH", 0)) / max(valu, 1), 3)
            if mean.get("SQ_ACTIVE_INST_VALU"):   # the profiler's VALUBusy: quad-cycles with a vector instruction in flight x 4 over the launch's SIMD-cycles
                e["valu_busy_frac"] = round(mean["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * avg * CLOCK_GHZ), 4)
            if "SQ_THREAD_CYCLES_VALU" in mean and "SQ_ACTIVE_INST_VALU" in mean and mean["SQ_ACTIVE_INST_VALU"]:
                # thread-cycles per active VALU quad-cycle: 64 lanes busy for a whole instruction = 16 per quad-cycle x 4
                e["valu_active_lanes_of_64"] = round(mean["SQ_THREAD_CYCLES_VALU"] / mean["SQ_ACTIVE_INST_VALU"], 1)
        wc = mean.get("SQ_WAVE_CYCLES")
        if wc:
            e["wave_wait_frac"] = round(mean.get("SQ_WAIT_ANY", 0) / wc, 3)
            e["wave_issue_stall_frac"] = round(mean.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
            e["wave_active_frac"] = round(mean.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
            if mean.get("GRBM_GUI_ACTIVE"):
                e["mean_resident_waves_per_simd"] = round(wc * 4 / (mean["GRBM_GUI_ACTIVE"] / 8) / SIMDS, 2)
            e["waves_per_launch"] = round(mean.get("SQ_WAVES", 0), 1)
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:   # KiB, separate passes; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
            b = mean["FETCH_SIZE"] * 1024 * 2 + mean["WRITE_SIZE"] * 1024
            e.update({"hbm_bytes_per_launch": int(b), "hbm_gb_per_s": round(b / avg, 1), "hbm_frac": round(b / avg / HBM_PEAK_GBS, 4)})
        if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean and mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]), 4)
        e["counters"] = {n: round(v, 1) for n, v in sorted(mean.items())}
        rows.append(e)
    trace_log = os.path.join(root, f"trace_{cfg}.log")
    frame_line = None
    if os.path.exists(trace_log):
        lines = [l for l in open(trace_log).read().splitlines() if "ms/frame" in l or l.startswith("{")]
        frame_line = lines[-1][:400] if lines else None
    return {"command": COMMANDS.get(cfg), "kernel_time_in_trace_ms": round(total_ns / 1e6, 3), "run_reported": frame_line,
            "dominant_kernel": rows[0]["kernel"] if rows else None, "kernels": rows}


def bench_counters(root, sha):
    """The C2 counters in the form bench.py reads (profiles/r03_pmc_counters.json)."""
    table = read_counters(os.path.join(root, "pmc_C2"))
    best = {}
    for k, c in table.items():
        if not k.startswith("k_"):
            continue
        base = re.sub(r"<.*", "", k)
        n = max(len(v) for v in c.values())
        if base not in best or best[base][0] < n:
            best[base] = (n, k, c)
    frames = best.get("k_generate", (1,))[0]
    res = {}
    for base, (n, k, c) in best.items():
        e = {name: sum(v) / len(v) for name, v in c.items()}
        e["variant"], e["dispatches"], e["launches_per_frame"] = k, n, max(1, round(n / max(frames, 1)))
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["traffic_bytes_per_launch"] = int(e["FETCH_SIZE"] * 1024 * 2 + e["WRITE_SIZE"] * 1024)
        res[base] = e
    near = [b for b in res if "intersect" in b or "camera" in b]
    shad = [b for b in res if "shadow" in b]
    pick = lambda names: max(names, key=lambda b: res[b].get("SQ_INSTS_VALU", 0) * res[b]["launches_per_frame"]) if names else None
    return {"source_sha": sha, "workload": "bench.py default frame (C2, 1920x1080), RT_LANES=1, one frame in flight: every kernel alone on the chip",
            "collected_with": "tools/evidence_r03.sh pmc (rocprofv3 --pmc, one run per counter set)", "nearest_kernel": pick(near), "shadow_kernel": pick(shad), "kernels": res}


def main(root):
    from raytracer_amd import buildinfo
    sha = buildinfo.source_sha()
    doc = {"source_sha": sha, "peaks": {"valu_issue_g_wave_instr_per_s": VALU_PEAK_GIPS, "hbm_gb_per_s": HBM_PEAK_GBS,
                                          "valu_issue_is": f"{SIMDS} SIMDs x {CLOCK_GHZ} GHz / {VALU_ISSUE_CYCLES:g} cycles per wave64 VALU instruction"},
           "collected_with": "tools/evidence_r03.sh pmc (RT_LANES=1: one stream, one frame at a time, every kernel alone on the chip)", "configs": {}}
    lines = [f"library sources {sha}; one MI355X; kernels alone (RT_LANES=1); frac = SQ_INSTS_VALU per launch / average launch time / {VALU_PEAK_GIPS:g} G wave-instr/s",
             "busy = SQ_ACTIVE_INST_VALU x 4 cycles / launch SIMD-cycles (the profiler's VALUBusy: about one quad-cycle booked per wave64 instruction, twice frac's 2 cycles)",
             "hbm = (FETCH_SIZE x 2 + WRITE_SIZE) per launch / launch time / 8 TB/s; L2 = TCC_HIT / (TCC_HIT + TCC_MISS); waves/SIMD = mean resident over the launch", ""]
    for cfg in ("C2", "C3", "C4", "C5"):
        if not os.path.isdir(os.path.join(root, f"pmc_{cfg}")):
            continue
        b = config_block(root, cfg)
        doc["configs"][cfg] = b
        lines.append(f"== {cfg}: {b['command']}")
        if b["run_reported"]:
            lines.append(f"   {b['run_reported'][:200]}")
        lines.append(f"{'kernel':52s} {'calls':>5s} {'share':>6s} {'avg us':>9s} {'VALU M':>8s} {'frac':>6s} {'busy':>6s} {'S+B/V':>6s} {'wait':>5s} {'HBM MB':>8s} {'hbm':>6s} {'L2 hit':>6s} {'waves/SIMD':>10s}")
        for e in b["kernels"]:
            lines.append(f"{e['kernel'][:52]:52s} {e['launches_in_trace']:5d} {e['share_of_kernel_time']:6.3f} {e['avg_launch_us']:9.1f} "
                         f"{e.get('valu_insts_per_launch', 0) / 1e6:8.2f} {e.get('frac_of_valu_issue', float('nan')):6.3f} {e.get('valu_busy_frac', float('nan')):6.3f} {e.get('salu_plus_branch_per_valu', float('nan')):6.2f} "
                         f"{e.get('wave_wait_frac', float('nan')):5.2f} {e.get('hbm_bytes_per_launch', 0) / 1e6:8.1f} {e.get('hbm_frac', float('nan')):6.3f} "
                         f"{e.get('l2_hit_rate', float('nan')):6.3f} {e.get('mean_resident_waves_per_simd', float('nan')):10.2f}")
        lines.append("")
    json.dump(doc, open(os.path.join(root, "config_roofline.json"), "w"), indent=1, sort_keys=True)
    open(os.path.join(root, "config_roofline.txt"), "w").write("\n".join(lines) + "\n")
    if os.path.isdir(os.path.join(root, "pmc_C2")):
        json.dump(bench_counters(root, sha), open(os.path.join(root, "pmc_counters.json"), "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03"))
