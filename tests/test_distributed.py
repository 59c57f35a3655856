"""N>1 path on CPU: world_size 2, gloo.  Covers the collectives the GPU path issues over
RCCL: one flat-gradient all-reduce per minibatch step, 3 tiny all-reduces per training step for the
normaliser, parameter broadcast at init, replica identity at the end."""
import json
import os
import subprocess
import sys

import helpers as H


def test_two_rank_training_keeps_replicas_identical():
    H.build_hostsim("float")  # build once, before the ranks race for it
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29611", os.path.join(H.ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    import re

    # the two ranks write to one pipe: their lines may run together
    res = [json.loads(m) for m in re.findall(r"RESULT (\{.*?\})(?=\s|RESULT|$)", p.stdout)]
    assert len(res) == 2
    for r in res:
        assert r["identical"]
        # normaliser saw the GLOBAL batch: 2 training steps x 8 envs x 4 steps
        assert r["count"] == 2 * 8 * 4
        # per training step: 2 normaliser all-reduces + per minibatch step TWO gradient all-reduces -- the value segment of
        # the flat buffer first (final when the value chain has joined; its exchange overlaps the policy network's backward
        # on a HIP device), then the policy segment -- each followed by Adam on that segment
        assert r["allreduce"] == 2 * (2 + 2 * 2 * 2)
        grad_sizes = [n for n in r["sizes"] if n in (r["n_value"], r["n_policy"])]
        assert grad_sizes == [r["n_value"], r["n_policy"]] * (2 * 2 * 2), grad_sizes
        assert r["n_value"] + r["n_policy"] == r["nparam"] or r["n_policy"] == r["nparam"]  # (params returned: the policy's)
