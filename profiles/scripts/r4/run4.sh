#!/usr/bin/env bash
# round 4, call 4: first voxel layer after the instruction-level pass (uniform rare branch, v_max3, packed dz fragments,
# branch-free Gram fragments, finalize fused): tests, stand-alone timings, step, rocprof kernel table of the step
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "l1 or volume or voxel or c2_ or bit_reproducible or aX or accumulator or trainer" > "$out/r4_t5.log" 2>&1; echo "rc=$?" >> "$out/r4_t5.log"
tail -8 "$out/r4_t5.log"
grep -q "rc=0" "$out/r4_t5.log" || exit 1
timeout -k 10 120 python3 tools/kbench.py l1 > "$out/r4_l1d.log" 2>&1; cat "$out/r4_l1c.log"
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 > "$out/r4_bench5.log" 2>&1; echo "rc=$?" >> "$out/r4_bench5.log"
python3 - "$out/r4_bench5.log" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l)
        print({k:d[k] for k in ("value","ms_per_step","value_with_input_transfer","step_mfma_frac")})
        print("family", d.get("roofline_family",{}).get("frac"), {k:round(v["ms"]*1e3,1) for k,v in d.get("roofline_family",{}).get("per_launch",{}).items()})
PY
bash profiles/run_prof.sh r04b > "$out/r4_prof5.log" 2>&1; head -60 "$out/prof_r04b.summary.txt"
