#!/bin/bash
# Round evidence in one go (run on the GPU box: `gpurun -- bash tools/collect_profiles.sh r02`).  Everything lands in
# gpurun_out/<tag>/; tools/summarise_profiles.py (run in the build container afterwards) turns it into profiles/<tag>_*.
# Counter passes carry --pmc only (no --kernel-trace / --stats with them); the program after `--` is python3 itself.
set -u
TAG=${1:-r05}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
W="python3 $ROOT/tools/run_workload.py"

echo "== bench lines"
$B > "$OUT/bench.json" 2> "$OUT/bench.err"
$B --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_shape.json" 2>> "$OUT/bench.err"
$B --obs-mode grid --no-cpu-baseline > "$OUT/bench_grid.json" 2>> "$OUT/bench.err"
$B --mixed --no-cpu-baseline > "$OUT/bench_mixed.json" 2>> "$OUT/bench.err"
$B --actions sweep --no-cpu-baseline > "$OUT/bench_sweep.json" 2>> "$OUT/bench.err"
$B --envs 32768 --steps 400 --warmup 50 --no-cpu-baseline > "$OUT/bench_n32768.json" 2>> "$OUT/bench.err"
$B --policy mlp --no-cpu-baseline > "$OUT/bench_mlp.json" 2>> "$OUT/bench.err"
$B --policy fragment --no-cpu-baseline > "$OUT/bench_fragment.json" 2>> "$OUT/bench.err"
$B --policy random-fragment --no-cpu-baseline > "$OUT/bench_random_fragment.json" 2>> "$OUT/bench.err"
$B --streams 2 --no-cpu-baseline > "$OUT/bench_streams2.json" 2>> "$OUT/bench.err"
$B --paint-method normal --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_normal.json" 2>> "$OUT/bench.err"
echo "== the reference's sheet: four mask words per lane (fine sheet), and with the stale kd-tree (coarse sheet)"
for p in square test; do $B --part $p --steps 600 --warmup 100 --no-cpu-baseline > "$OUT/bench_part_$p.json" 2>> "$OUT/bench.err"; done
echo "== large parts (Part_Dict rge:106-117: door_rr / door_rf / door_rr_big classes; step_kernel_big, mask rows in HBM)"
for spec in door_rr:328 door_rf:448 door_rr_big:652; do
  $B --part door_rr_big --tex ${spec##*:} --steps 300 --warmup 60 --no-cpu-baseline > "$OUT/bench_part_${spec%%:*}.json" 2>> "$OUT/bench.err"
done
$B --part door_rr_big --tex 652 --policy random-fragment --steps 300 --warmup 100 --no-cpu-baseline > "$OUT/bench_part_door_rr_big_random_fragment.json" 2>> "$OUT/bench.err"
echo "== cone beams on a large part (772 beams a shot at 70 654 samples: the beam table grows with the sample density)"
$B --paint-method normal --part door_rr_big --tex 652 --steps 60 --warmup 10 --no-cpu-baseline > "$OUT/bench_normal_part_door_rr_big.json" 2>> "$OUT/bench.err"
echo "== the atan2-sector observation (OBS_GRAD 6, bpw:1045-1061): door and a 70 411-sample part"
$B --obs-grad 6 --steps 600 --warmup 100 --no-cpu-baseline > "$OUT/bench_sectors6.json" 2>> "$OUT/bench.err"
$B --obs-grad 6 --part door_rr_big --tex 652 --steps 200 --warmup 40 --no-cpu-baseline > "$OUT/bench_sectors6_part_door_rr_big.json" 2>> "$OUT/bench.err"
echo "== COLOR_MODE 'HSI' (thickness bytes, bpw:384-434): the door, under the cone beams, a 70 654-sample part"
$B --color-mode HSI --steps 600 --warmup 100 --no-cpu-baseline > "$OUT/bench_hsi.json" 2>> "$OUT/bench.err"
$B --color-mode HSI --paint-method normal --steps 100 --warmup 20 --no-cpu-baseline > "$OUT/bench_hsi_normal.json" 2>> "$OUT/bench.err"
$B --color-mode HSI --part door_rr_big --tex 652 --steps 200 --warmup 40 --no-cpu-baseline > "$OUT/bench_hsi_part_door_rr_big.json" 2>> "$OUT/bench.err"
$B --color-mode HSI --policy random-fragment --steps 300 --warmup 100 --no-cpu-baseline > "$OUT/bench_hsi_random_fragment.json" 2>> "$OUT/bench.err"
$B --color-mode HSI --policy fragment --steps 300 --warmup 100 --no-cpu-baseline > "$OUT/bench_hsi_fragment.json" 2>> "$OUT/bench.err"
echo "== the RCCL path on one rank (PAINTRL_FORCE_DIST=1)"
PAINTRL_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29613 $B --steps 300 --warmup 50 --no-cpu-baseline > "$OUT/bench_rccl_world1.json" 2>> "$OUT/bench.err"
echo "== batch-size sweep (one wave per CU ... four per SIMD: the step's latency chain against its throughput)"
for n in 256 1024 2048 3072 4096; do
  $B --envs $n --steps 600 --warmup 100 --no-cpu-baseline > "$OUT/bench_envs$n.json" 2>> "$OUT/bench.err"
done
echo "== issue cost of the vector instruction classes (tools/microbench/valu_rate.hip)"
timeout -k 5 200 "$ROOT/tools/microbench/valu_rate" "$OUT/valu_rate.json" > "$OUT/valu_rate.txt" 2>&1
echo "== does a table stay in L2 from launch to launch (tools/microbench/l2_cold.hip)"
timeout -k 5 60 "$ROOT/tools/microbench/l2_cold" > "$OUT/l2_cold.txt" 2>&1
echo "== per-wave trace of back-to-back launches (a -DPRL_WAVE_TRACE build under tools/_ab/trace.so, if present)"
if [ -f "$ROOT/tools/_ab/trace.so" ]; then PRL_TRACE_B2B=12 PRL_TRACE_STEPS=30 timeout -k 10 200 python3 "$ROOT/tools/wave_trace.py" > "$OUT/wave_trace_b2b.txt" 2>&1; fi
echo "== kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- $B --no-cpu-baseline > "$OUT/trace.log" 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace_grid" --output-format csv -- $B --obs-mode grid --no-cpu-baseline > "$OUT/trace_grid.log" 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace_normal" --output-format csv -- $B --paint-method normal --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/trace_normal.log" 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace_big" --output-format csv -- $B --part door_rr_big --tex 652 --steps 200 --warmup 40 --no-cpu-baseline > "$OUT/trace_big.log" 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace_kd" --output-format csv -- $B --part test --steps 400 --warmup 100 --no-cpu-baseline > "$OUT/trace_kd.log" 2>&1
echo "== SQ / TA counters"
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH"
G2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH"
G3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
G4="TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
for MODE in section grid; do
  i=0
  for G in "$G1" "$G2" "$G3" "$G4"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $G -d "$OUT/pmc_${MODE}_$i" --output-format csv -- $W --obs-mode $MODE --steps 100 > "$OUT/pmc_${MODE}_$i.log" 2>&1 || echo "pmc $MODE pass $i failed"
  done
done
echo "== cone-beam step: counters of its beams kernel"
PMCONE="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
timeout -k 10 300 rocprofv3 --pmc $PMCONE -d "$OUT/pmc_cone_1" --output-format csv -- $W --paint-method normal --steps 40 > "$OUT/pmc_cone_1.log" 2>&1 || echo "pmc cone pass 1 failed"
timeout -k 10 300 rocprofv3 --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 -d "$OUT/pmc_cone_2" --output-format csv -- $W --paint-method normal --steps 40 > "$OUT/pmc_cone_2.log" 2>&1 || echo "pmc cone pass 2 failed"
echo "== HBM bytes (FETCH_SIZE and WRITE_SIZE in separate passes) + calibration on copy_mask_kernel"
for MODE in section grid; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/hbm_fetch_$MODE" --output-format csv -- $W --obs-mode $MODE --steps 200 > "$OUT/hbm_fetch_$MODE.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/hbm_write_$MODE" --output-format csv -- $W --obs-mode $MODE --steps 200 > "$OUT/hbm_write_$MODE.log" 2>&1
done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/cal_fetch" --output-format csv -- python3 "$ROOT/tools/hbm_calibration.py" > "$OUT/cal_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/cal_write" --output-format csv -- python3 "$ROOT/tools/hbm_calibration.py" > "$OUT/cal_write.log" 2>&1
# keep what is merged back small: counter CSVs of 100-200 dispatches are fine, the per-dispatch kernel trace is not
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
du -sh "$OUT"
