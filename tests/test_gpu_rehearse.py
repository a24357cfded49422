"""bench.py's N-rank path rehearsed on the one-GPU box, each time as a FRESH child process group (torch.distributed.run, two ranks,
gloo on host copies, both ranks rendering their interleaved bands on cuda:0): recipe W with the frame gathered to rank 0 and
checked bit for bit against the reference's golden, and config 5's recipe P (--samples --paths) through the same sharding,
min/max all-reduce, packing and gather. What a rehearsal prints is not a measurement; that it runs and what it assembles is."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rehearse(extra, ranks=2, timeout=420):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", str(ranks), "--rehearse", "--no-cpu"] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_rehearse_recipe_w_two_ranks():
    d = _rehearse(["--tag", "teapot2_240x135", "--steps", "6", "--warmup", "2", "--repeats", "2"])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0
    assert d["config"]["z_bit_exact_vs_reference_golden"] is True
    assert d["config"]["repeats"]["n"] == 2


def test_rehearse_recipe_p_two_ranks():
    """Config 5's mode through the N-rank path: every rank renders its bands of a path-traced frame."""
    d = _rehearse(["--tag", "p11_240x135", "--samples", "4", "--paths", "--steps", "2", "--warmup", "1", "--repeats", "1"])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "recipe P, 4 samples per pixel" in d["config"]["workload"]
