#!/bin/bash
# Runs the GPU parity suites against diagnostic builds in which every fast path is disabled in favour
# of its general counterpart (per-shot painting instead of the union pass, whole-table scans instead
# of ring searches, the general two-stage ray instead of the convex-neighbourhood path).  The
# product build never defines these macros.
set -e
cd "$(dirname "$0")/.."
for flags in "-DPRL_FORCE_PER_SHOT_PAINT" "-DPRL_FORCE_FULL_SCANS -DPRL_FORCE_GENERAL_RAY"; do
  out=$(mktemp -d)/libpaintrl_hip.so
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Iinclude -Ipaintrl_amd/csrc $flags \
        paintrl_amd/csrc/paintrl_hip.hip paintrl_amd/csrc/policy_mlp.hip -o "$out"
  echo "== $flags"
  PAINTRL_LIB="$out" python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q \
        -k "not missing_library" 2>&1 | tail -1
done
