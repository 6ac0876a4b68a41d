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
# the last quarter of each kernel's launches on their own (the env population drifts: late steps differ from early ones)
T=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 - "$T" > "$ROOT/gpurun_out/${TAG}_late_quarter.txt" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in sorted(rows.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    if len(v) < 100:
        continue
    v.sort()
    q = v[-len(v) // 4:]
    print(f"{k[:90]:90s} calls {len(v):6d}  all {sum(d for _, d in v) / len(v) / 1e3:8.1f} us  late quarter {sum(d for _, d in q) / len(q) / 1e3:8.1f} us")
    n = len(v)
    print("    by tenth of the run:", " ".join(f"{sum(d for _, d in v[i * n // 10:(i + 1) * n // 10]) / max(len(v[i * n // 10:(i + 1) * n // 10]), 1) / 1e3:.0f}" for i in range(10)))
PY
cat "$ROOT/gpurun_out/${TAG}_late_quarter.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
grep -o '"value": [0-9.]*' "$ROOT/gpurun_out/$TAG.log" | head -1
head -8 "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" | cut -c1-170
