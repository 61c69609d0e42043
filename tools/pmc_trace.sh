#!/bin/bash
# Instruction counters of the path kernels (separate --pmc pass, kernel-trace only): evidence for "issue bound".
# Usage: tools/pmc_trace.sh <tag>   -> gpurun_out/prof_<tag>/pmc_inst/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_inst -o pmc -- \
  python $REPO/tools/dev_gpu_check.py c3 > $OUT/pmc_inst.log 2>&1
tail -5 $OUT/pmc_inst.log
grep -h "trace_paths" $OUT/pmc_inst/pmc_counter_collection.csv | awk -F'","' '{print $9, $16, $17}' | cut -c1-60,200- | head -20
