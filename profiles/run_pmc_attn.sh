#!/usr/bin/env bash
# SQ counters of the three attention kernels stand-alone (tools/kbench.py attn: B=32, L=512, H=4, p = 0 / 0.1 / 0.3), one
# rocprofv3 --pmc pass per group; per-kernel means -> gpurun_out/pmc_attn_<tag>.summary.txt
# usage (GPU box, repo root): profiles/run_pmc_attn.sh <tag>
set -uo pipefail
tag="$1"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$root/gpurun_out/pmca_${tag}_$i" -- \
      python3 "$root/tools/kbench.py" attn > "$root/gpurun_out/pmca_${tag}_$i.log" 2>&1 || echo "group $i failed"
done
python3 - "$root" "$tag" <<'PY'
import csv, glob, sys, collections, re
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/gpurun_out/pmca_{tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(attn_\w+_kernel<[^>]*>)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k in sorted(acc):
    lines.append(k)
    for c, v in sorted(acc[k].items()):
        lines.append(f"    {c:28s} mean {sum(v) / len(v):.5g} over {len(v)} launches")
print("\n".join(lines))
open(f"{root}/gpurun_out/pmc_attn_{tag}.summary.txt", "w").write("\n".join(lines) + "\n")
PY
