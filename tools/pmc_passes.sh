#!/bin/bash
# Hardware-counter passes over a short benchmark run, one rocprofv3 invocation per pass (the tool does
# not split counters over passes; limits: 8 SQ, 4 TCC (FETCH_SIZE takes 3, WRITE_SIZE 2), 4 TCP, 2 TA).
# usage: tools/pmc_passes.sh OUTDIR WORKLOAD   (run on the GPU box from the repository root)
set -u
OUT=${1:-gpurun_out/pmc}
WL=${2:-bunny_1000}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# kernels one at a time, but in the form the default command runs them (frames in flight keep bounce launches in full waves)
export RTAMD_FULL_WAVES=1 RTAMD_FUSE=0
run() { # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- \
    python3 "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-seconds 0 --no-kernel-events --in-flight 1 --single-stream --workload "$WL" > "$ROOT/$OUT/$name.log" 2>&1 \
    || { echo "pass $name failed"; tail -3 "$ROOT/$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY &&
run sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_VALU_CVT SQ_BUSY_CU_CYCLES &&
run tc1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
