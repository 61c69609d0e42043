# developer probe: bench.py at several lane counts / settings   (bash tools/inflight_probe.sh under gpurun)
run() { # name, inflight, steps, env...
  name=$1; shift; f=$1; shift; k=$1; shift
  env "$@" timeout 280 python bench.py --no-cpu-baseline --inflight $f --steps $k --warmup 5 > gpurun_out/b_$name.json 2> gpurun_out/b_$name.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/b_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d["ms_per_step"], d["single_volume_ms"], d["skeletons"], d["sweep"]["fell_back_to_heap"], d["hbm_reserved_peak_gb"])
PY
}
run k20 4 20 A=1
run k20_stag 4 20 KIMI_BENCH_STAGGER=1
