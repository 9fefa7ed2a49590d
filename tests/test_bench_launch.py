"""bench.py --gpus N, started the way the driver starts --gpus 1 (plain `python bench.py ...`, no launcher):
the parent must spawn the ranks itself as child processes, relay rank 0's single JSON line and exit with the
children's code.  Runs here without a GPU in the dry mode (gloo, made-up records, no kernels): what is under
test is the launcher, the process group, the record gather and the one-line contract, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra_env=None):
    env = dict(os.environ, VSLAM_BENCH_DRY="1", VSLAM_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "0"],
                          env=env, capture_output=True, text=True, timeout=300)


def test_gpus2_self_launches_child_ranks_and_prints_one_line():
    r = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gather_ok"] is True and d["value"] is None
    assert "gloo" in d["config"]["parallelism"]


def test_world_size_mismatch_is_an_error_not_an_assert():
    # a launcher that started 1 rank while --gpus says 2
    r = _run(2, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999",
                 "VSLAM_BENCH_DRY": ""})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_under_the_drivers_launcher_stdout_is_one_json_line():
    """For N > 1 the driver starts the ranks itself (`python -m torch.distributed.run --nproc-per-node N bench.py
    --gpus N ...`).  Libraries write to stdout from C code while the process group comes up (gloo's connection notes
    here, RCCL's version banner on the GPU box); none of that may reach stdout."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, VSLAM_BENCH_DRY="1", VSLAM_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2, r.stdout


def test_the_drivers_scale_command_at_eight_ranks():
    """SCALE's last point exactly as the driver launches it — `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W` — rehearsed here
    with eight gloo ranks in the dry mode: the 8-way slices, the gather of 8 x P records and the per-rank timing fields
    of the line are what is under test."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, VSLAM_BENCH_DRY="1", VSLAM_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["gather_ok"] is True and d["steps"] == 2 and d["warmup"] == 1
    pr = d["per_rank_ms_per_step"]
    assert len(pr["ranks"]) == 8 and pr["min"] <= pr["max"]


def test_self_launch_ends_ranks_that_exceed_the_wall_clock_limit():
    """A rank stuck in a rendezvous or a collective must not hang the caller: the parent gives its ranks a wall-clock limit,
    ends the whole process group on expiry and exits non-zero with one line of reason (VSLAM_BENCH_DRY_SLEEP makes the dry
    ranks sleep)."""
    env = dict(os.environ, VSLAM_BENCH_DRY="1", VSLAM_BENCH_BACKEND="gloo", VSLAM_BENCH_DRY_SLEEP="60")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--launch-timeout", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 124 and "did not finish within 8 s" in r.stderr, (r.returncode, r.stderr[-500:])
    assert time.time() - t0 < 60 and not r.stdout.strip()


@pytest.mark.gpu
@pytest.mark.parametrize("comm", ["per-rank", "per-context"])
def test_the_n_gt_1_path_with_one_rccl_rank(comm):
    """The whole bench through the code path of N > 1 -- process group, the library's RCCL communicator(s), the record gather
    on the batches' streams, max-over-ranks timing -- with the single rank a one-GPU box allows (VSLAM_BENCH_FORCE_DIST=1)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, VSLAM_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-extras",
                        "--no-profile-pass", "--cpu-pairs", "8", "--cpu-all-cores-pairs", "0", "--pairs", "32", "--comm", comm],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    g = d["record_gather"]
    assert g["through"] == "rccl" and g["rccl_ranks"] == 1 and g["own_block_intact_on_every_rank"] is True
    assert g["communicators_per_rank"] == (1 if comm == "per-rank" else d["config"]["batches_in_flight"]) and g["comm"] == comm
    assert d["parity_in_bench"]["bit_exact"] is True and d["n_gpus"] == 1 and d["value"] > 0
