#!/usr/bin/env bash
# round 4, call 6: final first-voxel-layer kernels: tests, timings, PMC; first runs of the --config c4 / c5 bench legs
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "l1 or volume or voxel or c2_ or bit_reproducible or aX or accumulator or trainer" > "$out/r4_t6.log" 2>&1; echo "rc=$?" >> "$out/r4_t6.log"
tail -5 "$out/r4_t6.log"
grep -q "rc=0" "$out/r4_t6.log" || exit 1
timeout -k 10 120 python3 tools/kbench.py l1 > "$out/r4_l1e.log" 2>&1; cat "$out/r4_l1e.log"
for cfg in c5 c4; do
  timeout -k 10 600 python3 bench.py --config $cfg --steps 100 --warmup 10 > "$out/r4_bench_$cfg.log" 2>&1; echo "rc=$?" >> "$out/r4_bench_$cfg.log"
  tail -4 "$out/r4_bench_$cfg.log" | cut -c1-1500
done
timeout -k 10 900 bash profiles/run_pmc_kernels.sh r04_l1 pmcl1 l1_ > "$out/r4_pmc_l1b.log" 2>&1; tail -100 "$out/r4_pmc_l1b.log" | grep -v "^  SQ_INSTS_SALU\|SQ_INST_CYCLES\|ACTIVE_INST_SCA"
