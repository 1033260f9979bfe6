#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc csv output (one directory per pass, see tools/pmc_passes.sh)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def main(root):
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


if __name__ == "__main__":
    main(sys.argv[1])
