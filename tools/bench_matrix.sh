#!/bin/bash
# developer probe: the bench line's key figures under a few parking configurations (one gpurun call)
# usage: tools/bench_matrix.sh "<env assignments>" ...   e.g. tools/bench_matrix.sh "KH_PARK=0" "KH_PARK_PATIENCE=0.3"
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  out=gpurun_out/matrix_$i.json
  env $cfg timeout 400 python bench.py --steps 8 --warmup 1 --no-cpu-baseline > $out 2> gpurun_out/matrix_$i.err
  python - "$cfg" $out <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], {k: d[k] for k in ("value", "ms_per_step", "single_volume_ms", "hbm_reserved_peak_gb")}, "paths", d["phases_s"].get("paths"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
    print(open(sys.argv[2].replace(".json", ".err")).read()[-800:])
PY
done
