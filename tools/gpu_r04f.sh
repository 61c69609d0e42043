#!/bin/bash
# Round-4 GPU call F: the whole GPU suite, the bench lines for profiles/, kernel stats, counter passes (path kernel, EDT).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04f
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
echo "== full suite"; timeout 2400 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1; tail -8 $OUT/tests.txt
echo "== bench default"; timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json
echo "== bench driver"; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_s20.json 2> $OUT/bench_s20.err; python - <<PY
import json
for f in ("bench_default", "bench_s20"):
    try:
        d = json.load(open("$OUT/%s.json" % f)); print(f, d["value"], d["ms_per_step"], d["single_volume_ms"], d["volumes_in_flight"], d["hbm_reserved_peak_gb"], d.get("speedup_latency"), d.get("speedup_throughput"), d["roofline"]["frac"], d["roofline_trace"]["frac"])
    except Exception as e: print(f, "failed", e)
PY
echo "== kernel stats + EDT pmc"; bash tools/profile_round.sh r04 2>&1 | tail -3
echo "== path kernel pmc"; bash tools/pmc_trace_r3.sh r04 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
echo "== EDT SQ counters"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_r04/pmc_edt_sq -o pmc -- python $REPO/tools/edt_only.py c3 > $REPO/gpurun_out/prof_r04/pmc_edt_sq.log 2>&1; tail -1 $REPO/gpurun_out/prof_r04/pmc_edt_sq.log
