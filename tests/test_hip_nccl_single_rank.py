"""GPU: the multi-GPU path rehearsed on the one GPU a test box has — a one-rank process group on the
"nccl" backend (= RCCL) launched exactly as the scaling bench is (python -m torch.distributed.run), with the sharded
pipeline on the real WorldBatch backend and its small-result gather running as RCCL collectives on device tensors;
results must equal the unsharded batch bitwise.  (Two ranks cannot share one GPU under RCCL; world_size 2 and 4 are
covered on CPU by tests/test_distributed_gloo.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_batch_under_rccl_one_rank(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / "nccl_rank0.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(here, "_nccl_rank_script.py"), out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["dio"] is True and res["harvest"] is True
    assert res["max_over_ranks"] == 1.5


def _bench_two_ranks(extra, timeout=900):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", WH_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-extras", "--no-cpu-baseline", "--no-pmc"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line; rank 1 nothing
    return json.loads(lines[0])


def test_bench_two_ranks_time_sharing_one_gpu():
    """The command the driver runs for the scaling record, at N = 2, on the one GPU a test box has
    (WH_BENCH_SHARE_GPU=1: both ranks on device 0, gloo instead of RCCL): per-rank input generation, the barrier, the
    max / sum reductions and the gathered per-rank clocks of bench.py run with two processes driving GPU work at the
    same time.  Weak scaling (64 utterances per rank, shortened to 2 s here) and strong scaling (a fixed batch sharded
    by world.distributed.shard_ranges).  No scaling number is expected from this."""
    weak = _bench_two_ranks(["--seconds", "2"])
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["shared_gpu"] is True
    assert weak["steps"] == 2 and len(weak["per_rank_ms"]) == 2 and max(weak["per_rank_ms"]) == pytest.approx(weak["ms_per_step"], rel=1e-3)
    assert weak["device_flags"] == [0] * 16
    frames = 2 * 64 * 401  # both ranks' frames: int(1000 * 32000 / 16000 / 5 + 1) per utterance
    assert weak["value"] == pytest.approx(frames * 2 / (weak["ms_per_step"] * 2 / 1e3), rel=1e-6)
    strong = _bench_two_ranks(["--config", "4", "--utts", "6", "--seconds", "2", "--scaling", "strong"])
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong"
    assert strong["config"]["utterances_per_gpu"] == 3  # rank 0's share of the 6
    assert strong["value"] == pytest.approx(6 * 401 * 2 / (strong["ms_per_step"] * 2 / 1e3), rel=1e-6)
    assert strong["device_flags"] == [0] * 16
