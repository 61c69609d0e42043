#!/bin/bash
# ONE parametrised GPU-box script (run through gpurun); replaces the per-call tools/gpu_r0*.sh of rounds 4-5.
#
#   tools/gpu.sh <tag> <step> [<step> ...]        -> gpurun_out/<tag>/
#
# steps (each a single shell word; ':' separates its arguments):
#   gate[:file,file,...]         pytest -m gpu -x on the named test files (default: the trace / cube / ccl / edt / kat files)
#   full                         the whole GPU tier (pytest tests -m gpu)
#   single:<name>[:ENV=V,...]    one c3 volume alone on the GPU (bench.py --inflight 1 --steps 2), summary line
#   lanes:<name>:<steps>:<warmup>[:ENV=V,...]   the driver's call (bench.py --steps K --warmup W), summary line
#   bench:<name>:<args with , for spaces>[:ENV=V,...]   any other bench.py call
#   pmc:<name>[:ENV=V,...]       counter passes of the path kernel on one c3 volume (FETCH_SIZE / WRITE_SIZE / SQ / TCC, one set per run)
#   kstats:<name>[:ENV=V,...]    rocprofv3 --kernel-trace --stats of one bench step
#   ktrace:<name>:<bench args>[:ENV]   rocprofv3 --kernel-trace of a bench call, summarised as a timeline (tools/summarize_timeline.py)
#   probe:<name>:<threads>:LIB=<probe build>   per-phase cycles of the sweep's level loop (-DKH_SWEEP_PROBE build)
#   edt:<name>[:ENV=V,...]       the three EDT passes timed on the c3 volume (tools/edt_time.py)
#   edtpmc:<name>[:ENV=V,...]    SQ counter passes of the EDT kernels (tools/summarize_edt_pmc.py)
#   smoke                        __graft_entry__.smoke()
# ENV: e.g. LIB=build_variants/libkimi_base.so (another build of the library, loaded through KIMI_HIP_LIB), KH_TRACE_THREADS=64
set -u
TAG=${1:?tag}
shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
mkdir -p $KIMI_VOLUME_CACHE

envs() {   # "A=1,B=2" -> exports (LIB= is KIMI_HIP_LIB relative to the repo)
  local IFS=','
  for kv in ${1:-}; do
    case $kv in
      LIB=*) export KIMI_HIP_LIB=$REPO/${kv#LIB=} ;;
      *=*) export "$kv" ;;
    esac
  done
}

summary() {   # $1 = json file, $2 = name
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "value", d.get("value"), "ms/step", d.get("ms_per_step"), "single_ms", d.get("single_volume_ms"),
          "lanes", d.get("volumes_in_flight"), "hbm_gb", d.get("hbm_reserved_peak_gb"))
    print("   sweep", d.get("sweep"))
    ch = d.get("chains") or {}
    print("   sum_Mcyc", ch.get("sum_Mcyc"))
    for c in (ch.get("longest") or [])[:2]:
        print("   chain", c)
    print("   roofline", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("launches"))
    print("   phases", d.get("phases_s"))
except Exception as e:
    print(sys.argv[2], "failed:", e)
    try:
        print(open(sys.argv[1].replace(".json", ".err")).read()[-3000:])
    except Exception:
        pass
PY
}

for STEP in "$@"; do
  IFS=':' read -r KIND A1 A2 A3 A4 <<< "$STEP"
  echo "== $STEP"
  case $KIND in
    gate)
      FILES=${A1:-test_gpu_trace.py,test_gpu_cube.py,test_gpu_ccl.py,test_gpu_edt.py,test_gpu_kat.py,test_gpu_budget.py}
      ( envs "${A2:-}"; timeout 1500 python -m pytest $(echo $FILES | sed 's/[^,]*/tests\/&/g; s/,/ /g') -x -q -m gpu > $OUT/gate.txt 2>&1 )
      rc=$?; tail -4 $OUT/gate.txt
      if [ $rc -ne 0 ]; then echo "GATE FAILED"; grep -n "Error\|assert\|FAILED" $OUT/gate.txt | head -40; exit 1; fi ;;
    full)
      timeout 1800 python -m pytest tests -q -m gpu > $OUT/full.txt 2>&1; echo "rc=$?"; tail -6 $OUT/full.txt ;;
    single)
      ( envs "${A2:-}"; KIMI_BENCH_DUMP_TASKS=$OUT/${A1}_tasks.npz KIMI_BENCH_INFLIGHT=1 timeout 900 python bench.py --workload c3 --steps 2 --warmup 0 --no-cpu-baseline > $OUT/$A1.json 2> $OUT/$A1.err )
      summary $OUT/$A1.json $A1 ;;
    lanes)
      ( envs "${A4:-}"; timeout 1500 python bench.py --steps $A2 --warmup $A3 --no-cpu-baseline > $OUT/$A1.json 2> $OUT/$A1.err )
      summary $OUT/$A1.json $A1 ;;
    bench)
      ( envs "${A3:-}"; timeout 1500 python bench.py $(echo $A2 | tr ',' ' ') > $OUT/$A1.json 2> $OUT/$A1.err )
      summary $OUT/$A1.json $A1 ;;
    pmc)
      mkdir -p $REPO/gpurun_out/prof_$A1
      ( envs "${A2:-}"; cd /tmp && export TMPDIR=/tmp; i=0
        for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
          i=$((i+1))
          timeout 240 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_$A1/pmc_trace_$i -o pmc -- \
            python $REPO/tools/trace_only.py c3 > $REPO/gpurun_out/prof_$A1/pmc_trace_$i.log 2>&1
          echo "pass $i ($SET): rc=$?  $(grep TRACEONLY $REPO/gpurun_out/prof_$A1/pmc_trace_$i.log | tail -1)"
        done )
      python tools/summarize_trace_pmc.py $A1 $A1 || true ;;
    kstats)
      ( envs "${A2:-}"; cd /tmp && export TMPDIR=/tmp
        rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_$A1 -o kt -- \
          python $REPO/bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline > $OUT/kstats_$A1.json 2> $OUT/kstats_$A1.err )
      find $OUT/kstats_$A1 -name "*kernel_stats.csv" | head -1 | xargs -r head -8 ;;
    ktrace)  # ktrace:<name>:<bench args with , for spaces>[:ENV]   kernel timeline of a bench call (rocprofv3 --kernel-trace)
      ( envs "${A3:-}"; cd /tmp && export TMPDIR=/tmp
        rocprofv3 --kernel-trace --output-format csv -d $OUT/ktrace_$A1 -o kt -- \
          python $REPO/bench.py $(echo $A2 | tr ',' ' ') --no-cpu-baseline > $OUT/ktrace_$A1.json 2> $OUT/ktrace_$A1.err )
      python tools/summarize_timeline.py $(find $OUT/ktrace_$A1 -name "*kernel_trace.csv" | head -1) ;;
    probe)   # probe:<name>:<threads>[:ENV]  cycles per phase of the sweep's level loop (a -DKH_SWEEP_PROBE build given by LIB=)
      ( envs "${A3:-}"; KH_TRACE_THREADS=${A2:-64} timeout 600 python tools/trace_only.py c3 > $OUT/probe_$A1.txt 2>&1 )
      grep TRACEONLY $OUT/probe_$A1.txt
      python tools/summarize_probe.py $OUT/probe_$A1.txt ;;
    edt)     # edt:<name>[:ENV]   the three EDT passes on the c3 volume (tools/edt_time.py: kh_edt_timed, HIP events on the launch stream)
      ( envs "${A2:-}"; timeout 600 python tools/edt_time.py c3 > $OUT/edt_$A1.txt 2>&1 )
      grep EDTTIME $OUT/edt_$A1.txt || tail -5 $OUT/edt_$A1.txt ;;
    edtpmc)  # edtpmc:<name>[:ENV]   SQ counter passes of the EDT kernels on the c3 volume (tools/edt_only.py)
      mkdir -p $OUT/edtpmc_$A1
      ( envs "${A2:-}"; cd /tmp && export TMPDIR=/tmp; i=0
        for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
                   "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
          i=$((i+1))
          timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/edtpmc_$A1/pass_$i -o pmc -- \
            python $REPO/tools/edt_only.py c3 > $OUT/edtpmc_$A1/pass_$i.log 2>&1
          echo "pass $i: rc=$?"
        done )
      python tools/summarize_edt_pmc.py $OUT/edtpmc_$A1 ;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
    *) echo "unknown step $STEP" ;;
  esac
done
