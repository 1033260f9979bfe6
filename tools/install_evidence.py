#!/usr/bin/env python3
"""Copies what tools/pmc_r02.sh and the kernel-trace run left under gpurun_out/ into profiles/ (counter file, counter summary, per-kernel
table): python tools/install_evidence.py"""
import csv, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
import sys
sys.path.insert(0, ROOT)
from raytracer_amd import buildinfo
d = json.load(open("gpurun_out/r02_pmc/pmc_counters.json"))
assert d.get("source_sha") == buildinfo.source_sha(), (d.get("source_sha"), buildinfo.source_sha())
shutil.copy("gpurun_out/r02_pmc/pmc_counters.json", "profiles/r02_pmc_counters.json")
shutil.copy("gpurun_out/r02_pmc/summary.txt", "profiles/r02_pmc_summary.txt")
rows = list(csv.DictReader(open("gpurun_out/r02_trace/bench_kernel_stats.csv")))
with open("profiles/r02_bench_c2_kernel_stats.txt", "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline --no-baseline-configs   (RT_LANES=1: one stream, kernels run alone)\n")
    f.write(f"library sources {d['source_sha']} (profiles/r02_pmc_counters.json carries the same hash); one MI355X\n\n")
    f.write(f"{'kernel':70s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'min us':>9s} {'max us':>9s} {'%':>6s}\n")
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        f.write(f"{n[:70]:70s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.2f} {float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f} {float(r['Percentage']):6.2f}\n")
print(open("profiles/r02_bench_c2_kernel_stats.txt").read())
