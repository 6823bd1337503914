#!/usr/bin/env bash
# round 4, call 3: the restructured first voxel layer (Gram-matrix statistics, prefetched halo, backward without its second
# MFMA product): kernel + model tests first, stand-alone timings, then the step
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "l1 or volume or voxel or c2_ or bit_reproducible or aX or smoke or accumulator or trainer" > "$out/r4_t3.log" 2>&1; echo "rc=$?" >> "$out/r4_t3.log"
tail -15 "$out/r4_t3.log"
grep -q "rc=0" "$out/r4_t3.log" || exit 1
timeout -k 10 120 python3 tools/kbench.py l1 > "$out/r4_l1b.log" 2>&1; cat "$out/r4_l1b.log"
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 > "$out/r4_bench3.log" 2>&1; echo "rc=$?" >> "$out/r4_bench3.log"
python3 - "$out/r4_bench3.log" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l)
        print({k:d[k] for k in ("value","ms_per_step","value_with_input_transfer","step_mfma_frac")})
        print("roofline", {k:d["roofline"][k] for k in ("frac","avg_launch_ms","frac_minus_empty_bracket")})
        print("family", d.get("roofline_family",{}).get("frac"), {k:round(v["ms"]*1e3,1) for k,v in d.get("roofline_family",{}).get("per_launch",{}).items()})
        print("c4", d["roofline_c4"]["frac"], d["roofline_c4"]["avg_launch_ms"], "c2", d["roofline_c2_standalone"]["frac"])
PY
tail -3 "$out/r4_bench3.log" | cut -c1-300
