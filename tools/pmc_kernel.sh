#!/bin/bash
# rocprofv3 --pmc passes for ONE kernel of the library (substring KERNEL of its name), over tools/run_workload.py.
# Usage (on the GPU box): bash tools/pmc_kernel.sh TAG KERNEL [run_workload args]  ->  gpurun_out/pmc_TAG.txt
# Counters are collected in their own runs (no --kernel-trace / --stats with --pmc); every pass under `timeout`.
set -u
TAG=$1; KERNEL=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH"
G2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
G3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_RD"
G4="TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
G5="TA_FLAT_READ_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $G -d "$OUT/p$i" --output-format csv -- python3 "$ROOT/tools/run_workload.py" "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
PMC_KERNEL=$KERNEL python3 "$ROOT/tools/pmc_summary.py" "$OUT"/p* > "$ROOT/gpurun_out/pmc_$TAG.txt" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
cat "$ROOT/gpurun_out/pmc_$TAG.txt"
