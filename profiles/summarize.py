#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel-shape time per step."""
import collections
import csv
import sys


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:46]
        key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], r["Grid_Size_Z"])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault(key, []).append(d)
    tot = sum(sum(v) for v in agg.values())
    print(f"total kernel time {tot / 1e3:.2f} ms over {steps} steps = {tot / steps:.1f} us/step; launches/step = {len(rows) / steps:.0f}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:45]:
        print(f"{k[0]:46s} grid=({k[1]},{k[2]},{k[3]}) n/step={len(v) / steps:5.1f} avg={sum(v) / len(v):8.1f}us  per-step={sum(v) / steps:8.1f}us {100 * sum(v) / tot:5.1f}%")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 13)
