#!/bin/bash
# Bench lines of two prebuilt libraries side by side (GPU box): tools/ab_lines.sh tools/_ab/a.so tools/_ab/b.so
# (alternating, each line twice; prints steps/s)
cd "$(dirname "$0")/.."
line() { PAINTRL_LIB=$PWD/$1 python bench.py ${@:2} --no-cpu-baseline 2>/dev/null | python -c "import json,sys;print(round(json.loads(sys.stdin.read().splitlines()[-1])['value'],1))"; }
for cfg in "" "--obs-mode grid" "--mixed" "--policy fragment" "--policy random-fragment" "--policy mlp" "--paint-method normal --steps 200 --warmup 20"; do
  for rep in 1 2; do
    for lib in "$@"; do echo "$(basename $lib) [$cfg] $(line $lib $cfg)"; done
  done
done
for p in square test; do for lib in "$@"; do echo "$(basename $lib) part_$p $(PAINTRL_LIB=$PWD/$lib python bench.py --part $p --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200)"; done; done
