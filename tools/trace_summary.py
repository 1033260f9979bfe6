#!/usr/bin/env python3
"""Per-phase cycle counts from an RT_PROF step trace (RT_TRACE_FILE): usage  trace_summary.py trace.bin [--dump N]"""
import sys
import numpy as np

LEN = 2048
NAMES = {1: "packet start", 2: "rays+init", 3: "step start", 4: "staged", 5: "classified", 6: "step end", 7: "leaf start", 8: "leaf end",
         9: "walk done", 10: "fetched", 11: "stored", 12: "node read"}


def main(path, dump=0):
    raw = np.fromfile(path, dtype=np.uint64)[:2 * 4 * LEN].reshape(2, 4, LEN)
    for k, kname in enumerate(("nearest", "shadow")):
        for w in range(4):
            ev = raw[k, w]
            ev = ev[ev != 0]
            if len(ev) < 2:
                continue
            tag = (ev >> np.uint64(56)).astype(int)
            val = ((ev >> np.uint64(40)) & np.uint64(0xffff)).astype(int)
            t = (ev & np.uint64(0xffffffffff)).astype(np.int64)
            dt = np.diff(t)
            print(f"== {kname} wave slot {w}: {len(ev)} events, {t[-1] - t[0]} cycles, packets {np.sum(tag == 1)}")
            # attribute each interval to the (previous tag -> this tag) transition
            agg = {}
            for a, b, d in zip(tag[:-1], tag[1:], dt):
                key = (a, b)
                c = agg.setdefault(key, [0, 0])
                c[0] += 1; c[1] += int(d)
            for (a, b), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"   {NAMES.get(a, a):>13s} -> {NAMES.get(b, b):<13s} n={n:5d} total={tot:9d} avg={tot / n:8.0f}")
            if dump:
                for i in range(min(dump, len(ev))):
                    print(f"      {i:4d} {NAMES.get(tag[i], tag[i]):13s} val={val[i]:5d} t={t[i] - t[0]:8d} (+{dt[i - 1] if i else 0})")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
