#!/bin/bash
# Round-4 GPU call J: the lanes' EDF launches taking turns -- A/B at the driver's call.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04j
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
for cfg in "gate_on 1" "gate_off 0" "gate_on2 1"; do
  set -- $cfg
  echo "== $1"; KIMI_LANES_EDF_GATE=$2 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$1.json")); print("$1", d["value"], d["ms_per_step"], d["single_volume_ms"], d["volumes_in_flight"], d["hbm_reserved_peak_gb"])
except Exception as e: print("$1 failed", e); print(open("$OUT/$1.err").read()[-1500:])
PY
done
