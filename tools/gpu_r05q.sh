#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export KIMI_VOLUME_CACHE=/tmp/kimi_vol
timeout 300 python tools/edt_time.py c3 2>&1 | tail -1
for u in 8 14 28; do echo "unroll $u"; KIMI_HIP_LIB=$PWD/kimimaro_amd/libkimi_hip_u$u.so timeout 300 python tools/edt_time.py c3 2>&1 | tail -1; done
