"""The exact launch form the SCALE driver uses — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` — run end to end once on the one GPU a test box has
(N = 1; the launcher starts before anything touches the GPU).  Under a launcher bench.py creates the RCCL process group and
runs the data-parallel CVM_KITTI training leg (BASELINE configs[3]); CCVPE_ALLREDUCE_SINGLE_RANK=1 makes the 1-rank group
issue the real in-place ncclAvg all-reduces of the gradient arena.  Asserted: the JSON line's config.collective fields and
the compact config.train_dp_kitti_b64 summary; and that a failure of that leg still prints the line, then exits non-zero."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, batch):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CCVPE_ALLREDUCE_SINGLE_RANK="1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", str(batch), "--legs", "dp", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


def test_bench_under_torchrun_reports_the_collective():
    res, line = _run({}, 4)
    assert res.returncode == 0 and line is not None, res.stdout[-2000:] + res.stderr[-4000:]
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0
    assert len(json.dumps(line)) < 6000, "the JSON line must stay compact (per-kernel tables go to stderr)"
    assert "[bench kernels]" in res.stderr
    coll = line["config"]["collective"]
    assert coll["backend"].startswith("nccl") and coll["world_size_env"] == 1 and coll["ranks_counted_by_allreduce"] == 1
    assert coll["allreduce_calls_per_step"] == 3.0                       # the three gradient groups of the arena
    assert coll["bytes_per_step"] >= 4 * 55_000_000                      # CVM_KITTI: 60.4 M parameters (57.9 M with a gradient), fp32
    dp = line["config"]["train_dp_kitti_b64"]
    assert "error" not in dp and dp["ms_per_step"] > 0 and dp["pairs_per_s"] > 0 and dp["whole_step_frac"] > 0
    assert dp["loss_after_last_step"] == dp["loss_after_last_step"]      # finite


def test_bench_dp_failure_prints_the_line_and_exits_nonzero():
    res, line = _run({"CCVPE_BENCH_FAIL_DP": "1"}, 2)
    assert res.returncode != 0, "a failed data-parallel leg must fail the run"
    assert line is not None and line["value"] > 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "injected failure" in line["config"]["train_dp_kitti_b64"]["error"]
