#!/bin/bash
# Per-kernel durations of a bench.py invocation (rocprofv3 --kernel-trace --stats), on the GPU box:
#   bash tools/prof_kernels.sh TAG [bench.py args]   ->  gpurun_out/TAG_kernel_stats.csv (+ the bench line in TAG.log)
# PAINTRL_LIB selects another build.  Always under `timeout`.
set -u
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT" --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$ROOT/gpurun_out/$TAG.log" 2>&1
echo "rc=$?"
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
cp "$F" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
find "$OUT" -name "*kernel_trace.csv" -delete
grep -o '"value": [0-9.]*' "$ROOT/gpurun_out/$TAG.log" | head -1
head -8 "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" | cut -c1-170
