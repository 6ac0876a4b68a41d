#!/bin/bash
# Instruction counts of the step kernel's phases: one rocprofv3 --pmc pass per PRL_CUT build (prl_diag.hpp), the
# differences between consecutive builds are the phases.  Built beforehand in the build container:
#   python tools/build_variant.py cut0; for k in 1 2 4 5 6; do python tools/build_variant.py cut$k -DPRL_CUT=$k; done
# Usage (on the GPU box): bash tools/pmc_cuts.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_cuts
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
G="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES"
for k in 0 1 2 4 5 6; do
  export PAINTRL_LIB=$ROOT/tools/_ab/cut$k.so PAINTRL_LAX_SYMBOLS=1
  rocprofv3 --pmc $G -d "$OUT/c$k" --output-format csv -- python3 "$ROOT/tools/run_workload.py" --steps 60 > "$OUT/c$k.log" 2>&1 || echo "cut $k failed (see $OUT/c$k.log)"
  echo "== PRL_CUT=$k" >> "$OUT/summary.txt"
  python3 "$ROOT/tools/pmc_summary.py" "$OUT/c$k" >> "$OUT/summary.txt" 2>&1
done
find "$OUT" -name "*.csv" -size +2M -delete
cp "$OUT/summary.txt" "$ROOT/gpurun_out/pmc_cuts.txt"
cat "$ROOT/gpurun_out/pmc_cuts.txt"
