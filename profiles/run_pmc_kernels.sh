#!/usr/bin/env bash
# SQ / TCC counters of every kernel whose name contains <substr>, one rocprofv3 pass per counter group (FETCH_SIZE and
# WRITE_SIZE cannot share a pass on gfx950; MI355X_MICROARCH.md "rocprofv3 PMC slots"), reported PER KERNEL NAME
# (template instances apart).  The program follows `--` directly (no env / bash -c hop: the profiler has initialised the GPU).
# usage (GPU box, repo root): profiles/run_pmc_kernels.sh <tag> <tools/kbench.py case> <kernel name substring>
set -uo pipefail
tag="$1"; kcase="$2"; kname="$3"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  kt=""; [ $i -eq 2 ] && kt="--kernel-trace"          # one pass also keeps the (serialised) kernel durations
  rocprofv3 $kt --pmc $grp --output-format csv -d "$root/gpurun_out/pmck_${tag}_$i" -- \
      python3 "$root/tools/kbench.py" "$kcase" > "$root/gpurun_out/pmck_${tag}_$i.log" 2>&1 || echo "group $i ($grp) failed"
done
python3 - "$root" "$tag" "$kname" <<'PY'
import csv, glob, sys, collections, re
root, tag, kname = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/gpurun_out/pmck_{tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{root}/gpurun_out/pmck_{tag}_*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = []
for k in sorted(acc):
    lines.append(f"== {k}")
    c = acc[k]
    for n, v in sorted(c.items()):
        lines.append(f"  {n}: mean {sum(v)/len(v):.5g} over {len(v)} launches")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fe = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024 * 2     # KiB; gfx950: wide coalesced reads are tallied at 1/2
        wr = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024
        lines.append(f"  HBM traffic per launch: fetch (x2 gfx950 correction: an UPPER bound for narrow loads) {fe/1e6:.2f} MB "
                     f"[uncorrected {fe/2e6:.2f}] + write {wr/1e6:.2f} MB = {(fe+wr)/1e6:.2f} MB")
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if k in dur:
        d = sorted(dur[k])
        lines.append(f"  duration with the profiler serialising kernels: median {d[len(d) // 2]:.1f} us over {len(d)} launches")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            lines.append(f"  -> {(fe + wr) / 1e6 / d[len(d) // 2]:.2f} TB/s of ~8 (upper bound, see the fetch correction)")
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m:
        lines.append(f"  waiting fraction of wave cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES): {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m:
        lines.append(f"  VALU issue share (SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES): {m['SQ_ACTIVE_INST_VALU'] / m['SQ_BUSY_CYCLES']:.3f}")
print("\n".join(lines))
open(f"{root}/gpurun_out/pmck_{tag}.summary.txt", "w").write("\n".join(lines) + "\n")
PY
