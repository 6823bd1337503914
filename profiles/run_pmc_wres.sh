#!/usr/bin/env bash
# SQ / TCC counters of the weight-resident layer-2 conv3d kernel (csrc/conv3d_wres.hip), one pass per group
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage (GPU box, repo root): profiles/run_pmc_wres.sh <tag> <kbench case: pmc3d | pmc4 | pmcs> [kernel name substring]
set -uo pipefail
tag="$1"; kcase="${2:-pmc3d}"; kname="${3:-conv3d_wres}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$root/gpurun_out/pmcw_${tag}_$i" -- \
      python3 "$root/tools/kbench.py" "$kcase" > "$root/gpurun_out/pmcw_${tag}_$i.log" 2>&1 || echo "group $i failed"
done
python3 - "$root" "$tag" "$kname" <<'PY'
import csv, glob, sys, collections
root, tag, kname = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for f in glob.glob(f"{root}/gpurun_out/pmcw_{tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = [f"{k}: mean {sum(v)/len(v):.5g} over {len(v)} launches" for k, v in sorted(acc.items())]
if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
    fe = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"]) * 1024 * 2     # KiB; gfx950: wide coalesced reads are tallied at 1/2
    wr = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"]) * 1024
    lines.append(f"HBM traffic per launch: fetch (x2 gfx950 correction) {fe/1e6:.2f} MB + write {wr/1e6:.2f} MB = {(fe+wr)/1e6:.2f} MB")
print("\n".join(lines))
open(f"{root}/gpurun_out/pmcw_{tag}.summary.txt", "w").write("\n".join(lines) + "\n")
PY
