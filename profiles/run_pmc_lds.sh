#!/usr/bin/env bash
# LDS bank-conflict cycles vs LDS active cycles for every kernel of the training step.
# usage (GPU box, repo root): profiles/run_pmc_lds.sh <tag>
set -euo pipefail
tag="$1"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$root/gpurun_out/pmclds_$tag" -- \
    python3 "$root/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --profile > "$root/gpurun_out/pmclds_$tag.log" 2>&1
python3 - "$root" "$tag" <<'PY'
import csv, glob, sys, collections, re
root, tag = sys.argv[1], sys.argv[2]
f = glob.glob(f"{root}/gpurun_out/pmclds_{tag}/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60] + f" grid={r['Grid_Size']}"
    e = acc[name]
    if r["Counter_Name"] == "SQ_LDS_BANK_CONFLICT":
        e[0] += float(r["Counter_Value"]); e[2] += 1
    elif r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
        e[1] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1][0])
lines = [f"{'kernel':78s} {'conflict cyc':>13s} {'active cyc':>13s} {'ratio':>6s} {'launches':>8s}"]
for k, (c, a, n) in rows:
    if a > 0:
        lines.append(f"{k:78s} {c:13.3g} {a:13.3g} {c / a:6.2f} {n:8d}")
print("\n".join(lines[:40]))
open(f"{root}/gpurun_out/pmclds_{tag}.summary.txt", "w").write("\n".join(lines) + "\n")
PY
