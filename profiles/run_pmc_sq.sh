#!/usr/bin/env bash
# SQ counters of the layer-2 conv3d forward kernel (LDS conflicts, MFMA busy), one pass per group.
# usage (GPU box, repo root): profiles/run_pmc_sq.sh <tag>
set -euo pipefail
tag="$1"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$root/gpurun_out/pmcsq_${tag}_$i" -- \
      python3 "$root/tools/kbench.py" pmc3d > "$root/gpurun_out/pmcsq_${tag}_$i.log" 2>&1 || echo "group $i failed"
done
python3 - "$root" "$tag" <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"{root}/gpurun_out/pmcsq_{tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv3d_fwd_wres" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = [f"{k}: mean {sum(v)/len(v):.4g} over {len(v)} launches" for k, v in sorted(acc.items())]
print("\n".join(lines))
open(f"{root}/gpurun_out/pmcsq_{tag}.summary.txt", "w").write("\n".join(lines) + "\n")
PY
