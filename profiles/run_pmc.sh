#!/usr/bin/env bash
# HBM traffic of the layer-2 conv3d forward kernel: FETCH_SIZE and WRITE_SIZE need separate
# passes on gfx950 (TCC has 4 counter slots; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage (GPU box, repo root): profiles/run_pmc.sh <tag>
set -euo pipefail
tag="$1"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$root/gpurun_out/pmc_${tag}_$c" -- \
      python3 "$root/tools/kbench.py" pmc3d > "$root/gpurun_out/pmc_${tag}_$c.log" 2>&1
done
python3 - "$root" "$tag" <<'PY'
import csv, glob, sys
root, tag = sys.argv[1], sys.argv[2]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/gpurun_out/pmc_{tag}_{c}/*/*counter_collection.csv")[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if "conv3d_fwd_wres" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c] = (sum(vals) / len(vals), len(vals))
print(out)
# units: KiB per dispatch; gfx950 correction: FETCH_SIZE reports 1/2 of wide coalesced reads
fetch = out["FETCH_SIZE"][0] * 1024 * 2
write = out["WRITE_SIZE"][0] * 1024
print(f"conv3d_fwd_wres per launch: fetch(corrected x2) {fetch/1e6:.1f} MB, write {write/1e6:.1f} MB, total {(fetch+write)/1e6:.1f} MB")
open(f"{root}/gpurun_out/pmc_{tag}.summary.txt", "w").write(
    f"conv3d_fwd_wres_kernel (L2 32->64 @16^3, B=32): FETCH_SIZE raw {out['FETCH_SIZE'][0]:.1f} KiB x2 (gfx950 correction) = {fetch/1e6:.2f} MB; "
    f"WRITE_SIZE {write/1e6:.2f} MB; total {(fetch+write)/1e6:.2f} MB per launch over {out['FETCH_SIZE'][1]} launches\n")
PY
