"""The N > 1 path of bench.py on a one-GPU box: two ranks (fresh child processes, gloo, both on cuda:0: PYNQS_BENCH_REHEARSAL=1) must
print ONE JSON line with n_gpus = 2, and what the ranks agree on -- the all-reduced moments of E_loc and the all-reduced gradient --
must equal what ONE rank computes on the concatenation of the two shards (the deterministic SAMPLE_SPACE step; the semi-stochastic
REDUCE step draws per rank and is only checked to run)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(args, ranks, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    if ranks == 1:
        cmd = [sys.executable, "bench.py", "--gpus", "1"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "bench.py", "--gpus", str(ranks)] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"exactly one JSON line expected on stdout, got {len(lines)}"
    return json.loads(lines[0])


def test_two_ranks_agree_with_one_rank_on_the_concatenated_shard():
    common = ["--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--workload", "fe2s2_vmc_step"]
    two = _run(common + ["--walkers", "512"], 2, {"PYNQS_BENCH_REHEARSAL": "1"})
    one = _run(common + ["--walkers", "1024"], 1)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == "weak"
    assert two["config"]["walkers_per_gpu"] == 512 and "REHEARSAL" in two["data"]
    a, b = two["check"], one["check"]
    np.testing.assert_allclose(a["mean_eloc"], b["mean_eloc"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a["var_eloc"], b["var_eloc"], rtol=1e-9)
    np.testing.assert_allclose(a["grad_l2"], b["grad_l2"], rtol=1e-9)
    np.testing.assert_allclose(a["grad_first"], b["grad_first"], rtol=1e-8, atol=1e-12)


def test_the_default_reduce_step_runs_on_two_ranks():
    out = _run(["--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--walkers", "256"], 2, {"PYNQS_BENCH_REHEARSAL": "1"})
    assert out["n_gpus"] == 2 and out["config"]["workload"] == "fe2s2_reduce_vmc_step" and out["config"]["eps_sample"] == 1000
    assert out["value"] > 0 and np.isfinite(out["check"]["grad_l2"]) and out["parity"]["exact_part_bit_exact"]


@pytest.mark.parametrize("ranks", [3, 4])
def test_strong_scaling_with_uneven_shards(ranks):
    """--scaling strong: 1027 walkers of the whole job split as the reference splits its unique samples (utils/distributed/comm.py:108-111:
    the first total % world ranks one walker longer; probabilities pre-scaled by the world size, vmc/sample.py:772) over 3 and 4 ranks
    (the GPU box admits six processes on its card, the test runner and the launcher among them; the 8-rank case is the driver's, on 8 GPUs) must reproduce the one-rank moments and
    gradient of the same 1027 walkers."""
    common = ["--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--workload", "fe2s2_vmc_step", "--scaling", "strong", "--total-walkers", "1027"]
    many = _run(common, ranks, {"PYNQS_BENCH_REHEARSAL": "1"})
    one = _run(common, 1)
    assert many["n_gpus"] == ranks and many["scaling"] == "strong" and one["scaling"] == "strong"
    assert many["config"]["walkers_per_gpu"] == 1027 // ranks + (1 if 1027 % ranks else 0)   # (rank 0 holds one of the longer shards)
    assert one["config"]["walkers_per_gpu"] == 1027
    ph = many["step_phases_gpu_ms"]
    assert ph["stats_allreduce_bytes"] == 32 and ph["grad_allreduce_bytes"] > 0
    a, b = many["check"], one["check"]
    np.testing.assert_allclose(a["mean_eloc"], b["mean_eloc"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a["var_eloc"], b["var_eloc"], rtol=1e-9)
    np.testing.assert_allclose(a["grad_l2"], b["grad_l2"], rtol=1e-9)
    np.testing.assert_allclose(a["grad_first"], b["grad_first"], rtol=1e-8, atol=1e-12)
