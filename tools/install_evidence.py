#!/usr/bin/env python3
"""Copies what tools/evidence_r03.sh left under gpurun_out/r03 into profiles/ (counter file bench.py reads, per-configuration roofline
figures, per-kernel trace tables, bench line, configuration table): python tools/install_evidence.py
Refuses counters that were taken with other sources than the tree's."""
import csv
import glob
import json
import os
import shutil
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from raytracer_amd import buildinfo
import config_roofline

SRC = "gpurun_out/r03"
sha = buildinfo.source_sha()
d = json.load(open(f"{SRC}/pmc_counters.json"))
assert d.get("source_sha") == sha, (d.get("source_sha"), sha)
r = json.load(open(f"{SRC}/config_roofline.json"))
assert r.get("source_sha") == sha, (r.get("source_sha"), sha)
shutil.copy(f"{SRC}/pmc_counters.json", "profiles/r03_pmc_counters.json")
shutil.copy(f"{SRC}/config_roofline.json", "profiles/r03_config_roofline.json")
shutil.copy(f"{SRC}/config_roofline.txt", "profiles/r03_config_roofline.txt")
with open("profiles/r03_kernel_stats.txt", "w") as f:
    f.write(f"rocprofv3 --kernel-trace --stats, RT_LANES=1 (one stream, kernels run alone); library sources {sha}; one MI355X\n")
    for cfg in ("C2", "C3", "C4", "C5"):
        files = glob.glob(f"{SRC}/trace_{cfg}/**/*kernel_stats.csv", recursive=True)
        if not files:
            continue
        f.write(f"\n== {cfg}: {config_roofline.COMMANDS[cfg]}\n")
        f.write(f"{'kernel':70s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'min us':>9s} {'max us':>9s} {'%':>6s}\n")
        for row in list(csv.DictReader(open(files[0])))[:12]:
            n = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            f.write(f"{n[:70]:70s} {row['Calls']:>6s} {float(row['TotalDurationNs']) / 1e6:10.3f} {float(row['AverageNs']) / 1e3:10.2f} {float(row['MinNs']) / 1e3:9.2f} "
                    f"{float(row['MaxNs']) / 1e3:9.2f} {float(row['Percentage']):6.2f}\n")
for src, dst in (("bench_c2.json", "r03_bench_c2.json"), ("configs.json", "r03_configs.json")):
    if os.path.exists(f"{SRC}/{src}"):
        shutil.copy(f"{SRC}/{src}", f"profiles/{dst}")
        print("installed", dst)
print(open("profiles/r03_config_roofline.txt").read())
