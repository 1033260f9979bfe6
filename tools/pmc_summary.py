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


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
