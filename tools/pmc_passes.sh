#!/bin/bash
# Collects rocprofv3 PMC passes for bench.py, one counter group per run (gfx950 slot limits:
# 8 SQ, 4 TCC with FETCH_SIZE=3 and WRITE_SIZE=2, 4 TCP, 2 GRBM). Usage: tools/pmc_passes.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for GROUP in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
             "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/p$i" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu "$@" > "$OUT/p$i.log" 2>&1
  rc=$?
  echo "pass $i [$GROUP] rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/p$i.log"; exit $rc; fi
done
python3 "$R/tools/pmc_summary.py" "$OUT" "$OUT/pmc_traffic.json" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
