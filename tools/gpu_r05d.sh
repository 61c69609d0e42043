#!/bin/bash
# Round-5 GPU call D: isolate the abort of call C -- each step in its own process.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r05d
mkdir -p $OUT
run() { name=$1; shift; echo "== $name"; ( "$@" ) > $OUT/$name.txt 2>&1; echo "   rc=$?"; grep -v "^  File\|^Extension modules\|pluggy\|_pytest" $OUT/$name.txt | tail -4; dmesg 2>/dev/null | tail -3; }
run ccl timeout 300 python -m pytest tests/test_gpu_ccl.py -x -q -m gpu
run a9_heap_only timeout 300 python -m pytest tests/test_gpu_trace.py -x -q -m gpu -k "invalidate_ball_reference_goldens"
run a9_graph timeout 300 python -m pytest tests/test_gpu_trace.py -x -q -m gpu -k "voxel_graph_reference"
KH_BIG_LDS_HEAP=0 KH_GHOSTS=0 run skel_small_noghost timeout 300 python -m pytest tests/test_gpu_trace.py -x -q -m gpu -k "test_skeletonize_matches_oracle"
KH_BIG_LDS_HEAP=1 KH_GHOSTS=0 run skel_big_noghost timeout 300 python -m pytest tests/test_gpu_trace.py -x -q -m gpu -k "test_skeletonize_matches_oracle"
KH_BIG_LDS_HEAP=0 KH_GHOSTS=1 run skel_small_ghost timeout 300 python -m pytest tests/test_gpu_trace.py -x -q -m gpu -k "test_skeletonize_matches_oracle"
KH_BIG_LDS_HEAP=1 KH_GHOSTS=1 run skel_big_ghost timeout 300 python -m pytest tests/test_gpu_trace.py -x -q -m gpu -k "test_skeletonize_matches_oracle"
run sweep_and_heap timeout 600 python -m pytest tests/test_gpu_trace.py -x -q -m gpu -k "sweep_and_heap_paths"
