#!/usr/bin/env bash
# Collects the round-4 evidence on an MI355X box (run from the repo root, writes gpurun_out/r04_*):
#   in-graph phase stamps, step-level kernel table (rocprofv3 --kernel-trace --stats of bench.py) + per-queue timeline,
#   rocprof kernel durations of the roofline kernel stand-alone at both shapes - the round-3 form (kbench pmc3d / pmc4:
#   rounds of 20 launches) AND sustained (sus2 / sus4: 1 200 / 400 back-to-back launches, the condition bench.py's
#   roofline_c2_standalone / roofline_c4 time it under) -, roofline-kernel counters at the C2 and config-#4 shapes.
# Plain timings first: a --pmc pass leaves the clocks in the profiling state for the rest of the call.
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"
mkdir -p "$out"
cd "$root"
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps 2>&1 | grep -v "^{" | grep -v amdgpu > "$out/r04_step_phase_stamps.txt"
{ python3 tools/kbench.py stream 2>&1 | grep conv3d; python3 tools/kbench.py wgrad3 2>&1 | grep wgrad3d; python3 tools/kbench.py l1 2>&1 | grep conv3d_l1; } > "$out/r04_conv3d_family_standalone.txt"
{ python3 tools/kbench.py sus2 2>&1 | grep sustained; python3 tools/kbench.py sus4 2>&1 | grep sustained; } > "$out/r04_wres_sustained_events.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r04_prof" -- \
    python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/r04_prof.log" 2>&1
f=$(ls "$out"/r04_prof/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r04_step_kernel_summary.txt"
cp "$(ls "$out"/r04_prof/*/*_kernel_stats.csv | head -1)" "$out/r04_step_kernel_stats.csv"
python3 "$root/tools/step_timeline.py" "$f" full > "$out/r04_step_timeline_under_rocprof.txt" 2>&1
for shape in pmc3d pmc4 sus2 sus4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r04_k_$shape" -- python3 "$root/tools/kbench.py" $shape > "$out/r04_k_$shape.log" 2>&1
  python3 - "$out" "$shape" <<'PY'
import csv, glob, statistics, sys
out, shape = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/r04_k_{shape}/*/*_kernel_trace.csv")[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "conv3d_wres" in r["Kernel_Name"]]
c2 = shape in ("pmc3d", "sus2")
gf = 14.4955 if c2 else 86.973
how = "rounds of 20 launches with a synchronise between rounds (tools/kbench.py %s)" % shape if shape.startswith("pmc") else \
      "%d back-to-back launches, no synchronise (tools/kbench.py %s)" % (len(d) - 10, shape)
line = (f"conv3d_wres_kernel stand-alone ({'C2: B=32 16^3' if c2 else 'config #4: B=32 32x32x24'}), rocprofv3 --kernel-trace, {how}: "
        f"{len(d)} launches, mean {statistics.mean(d):.2f} us, median {statistics.median(d):.2f} us, min {min(d):.2f} us -> "
        f"{gf / statistics.mean(d):.3f} PFLOP/s = {gf / statistics.mean(d) / 2.5:.3f} of 2.5 PF dense bf16 (mean)")
print(line)
open(f"{out}/r04_wres_standalone_{shape}.txt", "w").write(line + "\n")
PY
done
"$root/profiles/run_pmc_wres.sh" r04_c2 pmc3d > /dev/null 2>&1
"$root/profiles/run_pmc_wres.sh" r04_c4 pmc4 > /dev/null 2>&1
"$root/profiles/run_pmc_lds.sh" r04 > /dev/null 2>&1
echo done
