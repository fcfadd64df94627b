"""Tests that turn themselves on where MORE THAN ONE GPU is visible (the one-GPU boxes of the pool skip them): the multi-GPU path on the
real "nccl" backend -- RCCL over xGMI -- between real peers.  On one GPU the protocol is covered by gloo at world 2 / 3 / 8
(tests/test_distributed.py), by several contexts on one GPU (tests/test_ranges.py) and by RCCL with a world of one (tests/rccl_world1.py)."""
import json
import os
import subprocess
import sys

import pytest

from _common import ROOT


def _gpus():
    import torch
    return torch.cuda.device_count()          # (counts devices without initialising the GPU on this image)


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("BENCH_EMULATE", None)
    return env


@pytest.mark.gpu
def test_one_stream_over_every_visible_gpu_on_rccl_equals_the_oracle():
    n = min(_gpus(), 8)
    if n < 2:
        pytest.skip("one GPU visible: the nccl path between peers needs two")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "rccl_world2.py")], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ("rccl world-of-%d ok" % n) in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_bench_on_every_visible_gpu_runs_on_the_nccl_backend():
    """`bench.py --gpus N` as the driver starts it for the scaling curve, at a size that takes seconds: the line must say that RCCL carried the
    exchange, that every rank was seen, and both of its own checks must be green."""
    n = min(_gpus(), 8)
    if n < 2:
        pytest.skip("one GPU visible: the nccl path between peers needs two")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--mib", "256", "--steps", "2", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["ranks_seen"] == n and line["n_gpus"] == n, line
    assert line["checks"]["stream_inflates_to_input_crc"] is True and line["checks"]["sample_stream_equals_cpu_port"] is True, line["checks"]
    assert line["config"]["rc"] == 0 and line["value"] > 0
