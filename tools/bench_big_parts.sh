#!/bin/bash
# The large-part step (parts beyond 16 384 samples: seven of the reference's ten, Part_Dict rge:106-117) on synthetic parts of
# the sample counts of door_rr (17 891), door_rf (33 316) and door_rr_big (70 654): bench lines in bench.py's schema and the
# kernel's rocprofv3 statistics.  Run on the GPU box: `gpurun -- bash tools/bench_big_parts.sh r05 [suffix]`.
set -u
TAG=${1:-r05}
SUF=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${TAG}_big$SUF
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
for spec in door_rr:328 door_rf:448 door_rr_big:652; do
  name=${spec%%:*}; tex=${spec##*:}
  echo "== $name (texture $tex)"
  timeout -k 10 400 $B --part door_rr_big --tex $tex --steps 300 --warmup 60 --no-cpu-baseline > "$OUT/bench_part_$name.json" 2>> "$OUT/bench.err" || echo "bench $name failed"
  tail -c 400 "$OUT/bench_part_$name.json"; echo
done
echo "== kernel trace (70 654-sample class)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace_big" --output-format csv -- $B --part door_rr_big --tex 652 --steps 200 --warmup 40 --no-cpu-baseline > "$OUT/trace_big.log" 2>&1 || echo "trace failed"
find "$OUT" -name "*kernel_stats.csv" | head -3
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
du -sh "$OUT"
