#!/bin/bash
# A/B of library builds on the default bench (ball painter), one rocprofv3 kernel trace each (on the GPU box):
#   bash tools/ab_fast.sh product noprio ...     (names under tools/_ab/, `product` = the in-tree library)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
  if [ "$v" = product ]; then unset PAINTRL_LIB; else export PAINTRL_LIB=$ROOT/tools/_ab/$v.so; fi
  bash "$ROOT/tools/prof_kernels.sh" abf_$v --steps 1000 --warmup 100 > /dev/null 2>&1
  echo "== $v: $(grep -o '"value": [0-9.]*' "$ROOT/gpurun_out/abf_$v.log" | head -1)"
  grep -A1 "step_kernel" "$ROOT/gpurun_out/abf_${v}_late_quarter.txt" | grep -v "^--" | paste - - | sed -E 's/\(anonymous namespace\):://g; s/ +/ /g' | cut -c1-200
done
