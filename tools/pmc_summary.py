#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc csv output (one directory per pass, see tools/pmc_passes.sh)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def traffic(table, out):
    """HBM-side bytes per launch, as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE and WRITE_SIZE are KiB
    from separate passes; on gfx950 FETCH_SIZE tallies 128 B requests at 64 B, so it is doubled; WRITE_SIZE is exact."""
    res = {}
    for k, c in table.items():
        if not k.startswith("k_") or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        f, w = c["FETCH_SIZE"], c["WRITE_SIZE"]
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        base = re.sub(r"<.*", "", k)   # template variants of one kernel: keep the one launched most (the non-counting build)
        if base in res and res[base]["dispatches"] >= len(f):
            continue
        res[base] = {"variant": k, "dispatches": len(f), "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                  "traffic_bytes_per_launch": int(fk * 1024 * 2 + wk * 1024)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


def main(root, traffic_out=None):
    table = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values per dispatch
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            table[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f"# per-dispatch averages from {root}")
    for k in sorted(table, key=lambda k: -sum(table[k].get("SQ_WAVE_CYCLES", [0]))):
        if not k.startswith("k_"):
            continue
        print(f"\n{k}  ({max(len(v) for v in table[k].values())} dispatches)")
        for c in sorted(table[k]):
            v = table[k][c]
            print(f"    {c:32s} {sum(v) / len(v):18.1f}")
    if traffic_out:
        traffic(table, traffic_out)


def counters_json(root, out):
    """profiles/r02_pmc_counters.json: per-launch averages of every counter for each kernel of the frame (the variant launched
    most), HBM-side bytes per launch, launches per frame, and the sha of the sources the library was built from -- bench.py takes
    its roofline numerators from this file and refuses it for any other build."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from raytracer_amd import buildinfo
    table = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            table[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
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
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:   # KiB, separate passes; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
            e["traffic_bytes_per_launch"] = int(e["FETCH_SIZE"] * 1024 * 2 + e["WRITE_SIZE"] * 1024)
        res[base] = e
    near = [b for b in res if "intersect" in b or "camera" in b]
    shad = [b for b in res if "shadow" in b]
    pick = lambda names: max(names, key=lambda b: res[b].get("SQ_INSTS_VALU", 0) * res[b]["launches_per_frame"]) if names else None
    doc = {"source_sha": buildinfo.source_sha(), "workload": "bench.py default frame (C2, 1920x1080), RT_LANES=1, one frame in flight: every kernel alone on the chip",
           "collected_with": "tools/pmc_r02.sh (rocprofv3 --pmc, one run per counter set)", "nearest_kernel": pick(near), "shadow_kernel": pick(shad), "kernels": res}
    json.dump(doc, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--counters-json":
        counters_json(sys.argv[2], sys.argv[3])
        sys.exit(0)
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
