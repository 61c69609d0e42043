#!/bin/bash
# Counter passes of the path kernel (trace_paths_kernel) on the bench volume c3, one --pmc set per run, kernel-trace only
# (MI355X_MICROARCH.md: HBM bytes from FETCH_SIZE / WRITE_SIZE in separate passes; never with the sys / hip traces).
# Usage (through gpurun): tools/pmc_trace_r3.sh <tag>   -> gpurun_out/prof_<tag>/pmc_trace_*/
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 170 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_trace_$i -o pmc -- \
    python $REPO/tools/trace_only.py c3 > $OUT/pmc_trace_$i.log 2>&1
  echo "pass $i ($SET): rc=$?  $(grep TRACEONLY $OUT/pmc_trace_$i.log | tail -1)"
done
python $REPO/tools/summarize_trace_pmc.py $TAG || true
