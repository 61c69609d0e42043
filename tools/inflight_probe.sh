run() { # name, inflight, env...
  name=$1; shift; f=$1; shift
  env "$@" timeout 250 python bench.py --no-cpu-baseline --inflight $f --steps 6 > gpurun_out/b_$name.json 2> gpurun_out/b_$name.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/b_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d["ms_per_step"], d["single_volume_ms"], d["skeletons"], d["sweep"]["fell_back_to_heap"])
PY
}
run f1_l8k 1 KH_SWEEP_LDS_LEVELS=8192
run f3_l8k 3 KH_SWEEP_LDS_LEVELS=8192
run f4_l8k 4 KH_SWEEP_LDS_LEVELS=8192
run f4_l4k 4 KH_SWEEP_LDS_LEVELS=4096
