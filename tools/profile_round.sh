#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun):
#   kernel-trace + stats of one bench step, and the HBM PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs,
#   kernel-trace only, as MI355X_MICROARCH.md prescribes) of the EDT on the same volume.
# Usage: tools/profile_round.sh <tag> [nopmc]      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o kt -- \
  python $REPO/bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
[ "${2:-}" = "nopmc" ] && exit 0
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- \
    python $REPO/tools/edt_only.py c3 > $OUT/pmc_$C.log 2>&1
done
find $OUT -name "*.csv" | head -20
tail -c 600 $OUT/bench_under_rocprof.json
