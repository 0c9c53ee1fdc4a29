#!/bin/bash
# One counter pass (instruction counts, lanes, wave time) over a short benchmark run; summary per kernel on stdout.
# usage: tools/pmc_quick.sh OUTDIR [WORKLOAD] [extra bench args]     (run on the GPU box from the repository root)
OUT=${1:-gpurun_out/pmcq}; WL=${2:-bunny_1080p}; shift 2
ROOT=$(pwd); mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp RTAMD_FULL_WAVES=1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d "$ROOT/$OUT/sq1" -- \
  python3 "$ROOT/bench.py" --steps 3 --warmup 1 --repeats 1 --cpu-seconds 0 --no-kernel-events --in-flight 1 --workload "$WL" "$@" > "$ROOT/$OUT/sq1.log" 2>&1 || { tail -3 "$ROOT/$OUT/sq1.log"; exit 1; }
cd "$ROOT" && python3 tools/pmc_summary.py "$OUT/sq1" | python3 -c '
import sys
cur=None; rows={}
for line in sys.stdin:
    if not line.startswith(" "): cur=line.strip(); rows[cur]={}; continue
    p=line.split(); rows[cur][p[0]]=(float(p[2]), int(p[3].strip("(n=)")))
print("%-34s %5s %10s %6s %9s %9s %7s" % ("kernel","n","VALU","lanes","waves","wavecyc/w","wait%"))
for k,c in rows.items():
    if "SQ_INSTS_VALU" not in c: continue
    v=c["SQ_INSTS_VALU"][0]; n=c["SQ_INSTS_VALU"][1]
    print("%-34s %5d %10.0f %6.1f %9.0f %9.0f %7.0f" % (k[:34], n, v, c["SQ_THREAD_CYCLES_VALU"][0]/max(c["SQ_ACTIVE_INST_VALU"][0],1), c["SQ_WAVES"][0], 4*c["SQ_WAVE_CYCLES"][0]/max(c["SQ_WAVES"][0],1), 100*c["SQ_WAIT_ANY"][0]/max(c["SQ_WAVE_CYCLES"][0],1)))
'
