#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
for rep in 1 2 3; do
for s in "20 5" "200 20"; do
  set -- $s
  python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps', d['steps'], 'warmup', d['warmup'], d['ms_per_step'], d['value'], 'h2d', d['value_with_input_transfer'])"
done
done
