#!/bin/bash
# developer probe: the round-2 tree (git worktree _r2, built here) benched on the same box as a control
cd _r2 && timeout 400 python bench.py --steps 8 --warmup 1 --no-cpu-baseline > ../gpurun_out/ctl_r2.json 2> ../gpurun_out/ctl_r2.err
python - <<'PY'
import json
try:
    d = json.loads(open("../gpurun_out/ctl_r2.json").read().strip().splitlines()[-1])
    print("ROUND2-CONTROL", {k: d[k] for k in ("value", "ms_per_step", "single_volume_ms")}, "paths", d["phases_s"].get("paths"))
except Exception as e:
    print("control failed", e); print(open("../gpurun_out/ctl_r2.err").read()[-600:])
PY
