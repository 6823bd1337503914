#!/usr/bin/env bash
# Collects the round-3 evidence on an MI355X box (run from the repo root, writes gpurun_out/r03_*):
#   step-level kernel table (rocprofv3 --kernel-trace --stats of bench.py), per-queue timeline of one replayed step,
#   LDS bank-conflict table of the step, roofline-kernel counters at the C2 and config-#4 shapes, rocprof kernel
#   durations of the roofline kernel stand-alone at both shapes, in-kernel cycle stamps and the ablation sweep.
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"
mkdir -p "$out"
# plain timings first: a --pmc pass leaves the clocks in the profiling state for the rest of the call
cd "$root"
MMEEG_HIP_LIB="$root/multimodal_eeg_fmri_amd/csrc/build/abl_s0_n.so" python3 tools/kbench.py stamp 2>&1 | grep -v amdgpu > "$out/r03_wres_cycle_stamps.txt"
python3 tools/kbench.py c4b 2>&1 | grep conv3d > "$out/r03_wres_graph_replayed.txt"
{ python3 tools/kbench.py stream 2>&1 | grep conv3d; python3 tools/kbench.py wgrad3 2>&1 | grep wgrad3d; } > "$out/r03_conv3d_family_standalone.txt"
python3 tools/kbench.py attn 2>&1 | grep attention > "$out/r03_attention_standalone.txt"
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps 2>&1 | grep -v "^{" | grep -v amdgpu > "$out/r03_step_phase_stamps.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r03_prof" -- \
    python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/r03_prof.log" 2>&1
f=$(ls "$out"/r03_prof/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r03_step_kernel_summary.txt"
cp "$(ls "$out"/r03_prof/*/*_kernel_stats.csv | head -1)" "$out/r03_step_kernel_stats.csv"
python3 "$root/tools/step_timeline.py" "$f" full > "$out/r03_step_timeline_under_rocprof.txt" 2>&1
for shape in pmc3d pmc4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r03_k_$shape" -- python3 "$root/tools/kbench.py" $shape > "$out/r03_k_$shape.log" 2>&1
  python3 - "$out" "$shape" <<'PY'
import csv, glob, statistics, sys
out, shape = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/r03_k_{shape}/*/*_kernel_trace.csv")[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "conv3d_wres" in r["Kernel_Name"]]
gf = 14.4955 if shape == "pmc3d" else 86.973
line = (f"conv3d_wres_kernel stand-alone ({'C2: B=32 16^3' if shape == 'pmc3d' else 'config #4: B=32 32x32x24'}), rocprofv3 --kernel-trace: "
        f"{len(d)} launches, mean {statistics.mean(d):.2f} us, median {statistics.median(d):.2f} us, min {min(d):.2f} us -> "
        f"{gf / statistics.mean(d) * 1e3 / 1e3:.3f} PFLOP/s = {gf / statistics.mean(d) / 2.5:.3f} of 2.5 PF dense bf16 (mean)")
print(line)
open(f"{out}/r03_wres_standalone_{shape}.txt", "w").write(line + "\n")
PY
done
"$root/profiles/run_pmc_wres.sh" r03_c2 pmc3d > /dev/null 2>&1
"$root/profiles/run_pmc_wres.sh" r03_c4 pmc4 > /dev/null 2>&1
"$root/profiles/run_pmc_lds.sh" r03 > /dev/null 2>&1
echo done
