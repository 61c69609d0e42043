#!/bin/bash
# which wait removes the heap emulation's timing-dependent failure? (each variant has an iteration cap: no hang)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in w0 wA wB wC; do
  echo "== $v"; KIMI_HIP_LIB=$PWD/kimimaro_amd/libkimi_hip_$v.so SWEEP=0 timeout 60 python tools/selftest/a9_debug.py 2>&1 | grep -c " ok " ; KIMI_HIP_LIB=$PWD/kimimaro_amd/libkimi_hip_$v.so SWEEP=0 timeout 60 python tools/selftest/a9_debug.py 2>&1 | grep "BAD\|EXC\|^bad" | head -5
done
