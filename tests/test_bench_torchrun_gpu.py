"""The exact launch form the SCALE driver uses — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` — run end to end once on the one GPU a test box has
(N = 1; the launcher starts before anything touches the GPU).  Under a launcher bench.py creates the RCCL process group and
runs the data-parallel CVM_KITTI training leg (BASELINE configs[3]); CCVPE_ALLREDUCE_SINGLE_RANK=1 makes the 1-rank group
issue the real in-place ncclAvg all-reduces of the gradient arena.  Asserted: the JSON line's flat config.collective_* fields and
the flat config.train_dp_kitti_b64_* summary (at B = 64 per GPU: BASELINE configs[3]'s own batch); that a failure of that leg
still prints the line, then exits non-zero; and that a --gpus / WORLD_SIZE mismatch is refused."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, batch, gpus=1):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CCVPE_ALLREDUCE_SINGLE_RANK="1", CCVPE_BENCH_DP_STEPS="2,2", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
           "--batch", str(batch), "--legs", "dp", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


def test_bench_under_torchrun_reports_the_collective():
    res, line = _run({}, 64)
    assert res.returncode == 0 and line is not None, res.stdout[-2000:] + res.stderr[-4000:]
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0
    assert len(json.dumps(line)) < 6000, "the JSON line must stay compact (per-kernel tables go to stderr)"
    assert "[bench kernels]" in res.stderr
    cfg = line["config"]
    assert all(not isinstance(v, (dict, list)) for v in cfg.values()), "config must hold flat scalars only (the driver drops nested objects)"
    assert len([k for k, v in cfg.items() if not isinstance(v, (dict, list))]) <= 24, "the driver's record keeps 24 config scalars"
    assert cfg["collective_backend"].startswith("nccl") and cfg["collective_ranks"] == 1 and line["scaling"] == "weak"
    assert cfg["collective_calls_per_step"] == 3.0                       # the three gradient groups of the arena
    assert cfg["collective_bytes_per_step"] >= 4 * 55_000_000            # CVM_KITTI: 60.4 M parameters (57.9 M with a gradient), fp32
    assert "train_dp_kitti_b64_error" not in cfg, cfg.get("train_dp_kitti_b64_error")
    assert cfg["train_dp_kitti_b64_ms"] > 0 and cfg["train_dp_kitti_b64_pairs_per_s"] > 0 and cfg["train_dp_kitti_b64_frac"] > 0
    # what the 24-scalar record has no room for is on stderr, one JSON object per leg
    det = [json.loads(l[len("[bench leg] "):]) for l in res.stderr.splitlines() if l.startswith("[bench leg] ")]
    leg = [d["train_dp_kitti_b64"] for d in det if "train_dp_kitti_b64" in d][0]
    assert leg["batch_per_gpu"] == 64 and leg["loss_after_last_step"] == leg["loss_after_last_step"]      # finite
    assert leg["roofline"]["whole_step"]["work"] == "kitti"


def test_bench_dp_failure_prints_the_line_and_exits_nonzero():
    res, line = _run({"CCVPE_BENCH_FAIL_DP": "1"}, 2)
    assert res.returncode != 0, "a failed data-parallel leg must fail the run"
    assert line is not None and line["value"] > 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "injected failure" in line["config"]["train_dp_kitti_b64_error"]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    res, line = _run({}, 2, gpus=2)                 # one rank under the launcher, but --gpus 2
    assert res.returncode != 0 and line is None, res.stdout[-2000:]
    assert "WORLD_SIZE=1" in res.stderr
