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


def _run(extra, port):
    env = dict(os.environ, WAPT_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
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
    # no counter passes with more than one rank: the roofline object carries what the run itself measured
    assert d["roofline"]["bound"] == "valu" and d["roofline"]["grays_per_s_in_kernel"] > 0
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
