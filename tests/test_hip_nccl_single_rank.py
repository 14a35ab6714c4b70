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
