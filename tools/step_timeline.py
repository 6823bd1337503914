"""per-queue timeline of one hipGraph-replayed training step from a rocprofv3 kernel trace
usage: python tools/step_timeline.py <kernel_trace.csv> [full]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
idx = [i for i, r in enumerate(rows) if "adamw_finish" in r["Kernel_Name"]]
a, b = idx[-3] + 1, idx[-2] + 1
step = rows[a:b]
t0 = step[0]["s"]
print(f"step: {len(step)} kernels, span {(step[-1]['e'] - t0) / 1e3:.1f} us")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r["e"] - r["s"] for r in rs) / 1e3
    gaps = [(rs[i + 1]["s"] - rs[i]["e"]) / 1e3 for i in range(len(rs) - 1)]
    big = [(round((rs[i + 1]["s"] - t0) / 1e3), round(g)) for i, g in enumerate(gaps) if g > 20]
    print(f"queue {q}: n={len(rs)} busy={busy:.0f} us first={(rs[0]['s'] - t0) / 1e3:.0f} last={(rs[-1]['e'] - t0) / 1e3:.0f} "
          f"gaps>20us (at, len)={big}")
if len(sys.argv) > 2:
    for q, rs in sorted(byq.items()):
        print("--- queue", q)
        for i, r in enumerate(rs):
            gap = (r["s"] - rs[i - 1]["e"]) / 1e3 if i else 0
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:60]
            print(f"{(r['s'] - t0) / 1e3:7.1f} +{gap:5.1f} {(r['e'] - r['s']) / 1e3:6.1f} {name}")
