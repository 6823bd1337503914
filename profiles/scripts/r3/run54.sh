#!/usr/bin/env bash
# robustness: the GPU suite with every epilogue on its run-time (generic) form, and with the round's fusions switched off
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
MM_EPI_GENERIC=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$out/r3_t54a.log" 2>&1; echo "generic epilogues rc=$?"; tail -2 "$out/r3_t54a.log"
MM_NO_BNRED=1 MM_NO_GEMM2=1 MM_NO_BCAST=1 MM_NO_QKV_FUSE=1 MM_CONV_WGRADS_HANDED=2 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$out/r3_t54b.log" 2>&1; echo "fusions off rc=$?"; tail -2 "$out/r3_t54b.log"
MM_FFN1_FUSE=1 timeout -k 10 900 python3 -m pytest tests/test_models_gpu.py tests/test_trainer_gpu.py -x -q > "$out/r3_t54c.log" 2>&1; echo "ffn1 fuse rc=$?"; tail -2 "$out/r3_t54c.log"
