"""bench.py's N > 1 control flow on a one-GPU box: two ranks launched the way the driver launches them
(python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2), sharing the GPU, the skeleton exchange over gloo
(KIMI_BENCH_BACKEND=gloo; the measured configuration is nccl = RCCL, exercised at world size 1 in
tests/test_gpu_configs.py::test_all_gather_v_over_rccl_on_the_device).  Checks the JSON line's contract for both scaling
modes and that the per-rank times are reported."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_gloo_dry_run():
    env = dict(os.environ, KIMI_BENCH_BACKEND="gloo", KIMI_BENCH_INFLIGHT="2", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
           "--workload", "mini", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["unit"] == "labels/s"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["skeletons"] >= 30
    rt = d["rank_times"]
    assert len(rt["own_s_per_step"]) == 2 and len(rt["gather_s_per_step"]) == 2 and min(rt["own_s_per_step"]) > 0
    assert d["weak_scaling"]["scaling"] == "weak" and d["weak_scaling"]["skeletons"] >= 2 * 30   # both ranks' volumes gathered
    assert d["roofline"]["frac"] > 0 and d["roofline"]["bytes_per_launch"] > 0 and d["roofline_edt"]["frac"] > 0
