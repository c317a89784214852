"""bench.py's multi-rank control flow on a box with one GPU: two ranks under torch.distributed.run, both rendering on
cuda:0, the collectives over gloo through host copies (WAPT_BENCH_REHEARSAL=1).  The sequence of collectives is the one
the RCCL run issues - warm-up and timed frames with their reduce, the barrier-fenced timing, the max/sum all-reduces,
rank 0's profiled frames WITHOUT a collective, the closing barrier - so a rank that issues one the others do not
shows up here as a hang (the timeout) instead of on the 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port, ranks=2, env_extra=None):
    env = dict(os.environ, WAPT_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--res", "256x144", "--scene", "cornell", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=240)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_two_ranks_run_the_whole_bench_and_rank0_prints_one_line():
    d = _run(["--spp", "4"], 29541)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "8 spp image" in d["config"]["workload"]
    # no counter passes with more than one rank: the roofline object carries what the run itself measured (HIP events,
    # counting build); the counter-based fraction only where a committed profile of the same scene can be replayed
    assert d["roofline"]["grays_per_s_in_kernel"] > 0 and d["roofline"]["algorithmic"]["bytes_per_ray"] > 40
    assert d["roofline"]["counter_source"] is None and d["roofline"]["frac"] is None
    assert "cpu_baseline" not in d  # rank 0 at N=1 only


def test_two_ranks_share_one_image_in_the_strong_scaling_mode():
    """--total-spp: ONE image of that many samples, rank r renders sample_range(total, r, 2) - the sharding of the
    target (each GPU renders spp/N).  7 samples: the ranks get 4 and 3."""
    d = _run(["--total-spp", "7"], 29542)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "7 spp image sharded over 2 GPU(s)" in d["config"]["workload"]
    # paths of the whole job = W x H x 7 per step
    assert abs(d["value"] * d["ms_per_step"] * 1e3 - 256 * 144 * 7) <= 1e-6 * 256 * 144 * 7


def test_one_rank_through_rccl():
    """The collectives of the N > 1 flow over RCCL itself (backend "nccl"), with the one rank this box has a GPU for:
    process-group set-up on the device, reduce, barriers and all-reduces next to this library's own HIP stream."""
    env = dict(os.environ, WAPT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--res", "256x144", "--scene", "cornell", "--spp", "4", "--no-pmc", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=240)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0


def test_eight_ranks_render_the_image_of_one(tmp_path):
    """The target's sharding in full: ONE image of 1024 samples over 8 ranks (128 each), all eight on this box's one GPU
    with a queue budget of 4 GB per process, against the same image rendered by a single rank.  The samples are the
    same - a sample's random numbers follow from its global index - so the images agree to the order of float additions."""
    import numpy as np
    one, eight = str(tmp_path / "one.npy"), str(tmp_path / "eight.npy")
    env = {"WAPT_QUEUE_BUDGET_GB": "4"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--res", "256x144", "--scene", "cornell",
                        "--total-spp", "1024", "--no-pmc", "--no-cpu-baseline", "--no-roofline", "--dump-image", one],
                       capture_output=True, text=True, env=dict(os.environ, **env), cwd=ROOT, timeout=240)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _run(["--total-spp", "1024", "--no-roofline", "--dump-image", eight], 29544, ranks=8, env_extra=env)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and "128 spp each" in d["config"]["workload"]
    assert d["config"]["queue_bytes"] <= 4e9
    a, b = np.load(one), np.load(eight)
    assert a.shape == (144, 256, 3) and np.isfinite(a).all() and a.mean() > 0.05
    # (1024 float32 terms per pixel, added in another order: 99.9 % of the values agree to 1e-5, the rest to 6e-5)
    assert (np.abs(b - a) <= 1e-5 * np.abs(a) + 1e-6).mean() >= 0.995
    np.testing.assert_allclose(b, a, rtol=2e-4, atol=1e-6)
