"""The N > 1 leg of bench.py with KERNELS running: two ranks launched exactly as the driver launches them (torch.distributed.run,
one fresh process per rank, rendezvous on 127.0.0.1), sharing card 0 through VIEKF_DIST_BACKEND=gloo -- the rehearsal a one-GPU
box allows (r02 kept its record only in scratch).  Each rank steps its own shard of the filters on the device,
checks 8 of them against the oracle, and the run's one collective reduces {steps, seconds, bytes, max_rel_err}."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_card_step_their_shards_and_reduce_one_record():
    B, steps, warm = 96, 4, 1
    env = dict(os.environ, VIEKF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warm),
           "--batch", str(B), "--no-secondary"]     # (the parity leg stays on: every rank checks 8 of its filters)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == steps and j["warmup"] == warm and j["scaling"] == "weak"
    assert j["config"]["batch_per_gpu"] == B and "no collective" in j["config"]["parallelism"]
    # both ranks' filters against the oracle: the reduced maximum is inside the parity bar, nobody reported a flag
    assert j["parity_max_rel_err"] <= 1e-6 and j["nan_filters"] == 0
    red = j["reduction"]                                          # {sum, max, sum, max} over the ranks
    assert red["steps"] == 2 * B * steps and red["max_rel_err"] == j["parity_max_rel_err"]
    assert red["seconds"] > 0 and red["bytes"] > 0
    assert abs(j["value"] - red["steps"] / red["seconds"]) <= 1e-6 * j["value"]      # whole-job rate = all ranks' steps / max seconds
    assert abs(j["ms_per_step"] - 1e3 * red["seconds"] / steps) <= 1e-6 * j["ms_per_step"]
