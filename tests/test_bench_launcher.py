"""`python bench.py --gpus N` WITHOUT a launcher (the driver's N = 1 command form, extended to N > 1): the parent starts the N
ranks itself under torch.distributed.run — before anything touches a GPU — relays rank 0's line and exits with the
launcher's code.  Runs here on CPU with a stub as the program the ranks execute (CCVPE_BENCH_CHILD): the stub joins a gloo
group, counts the ranks with an all-reduce and prints a line on rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = '''
import argparse, json, os, sys
import torch, torch.distributed as dist
ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int); ap.add_argument("--steps", type=int); ap.add_argument("--warmup", type=int)
a = ap.parse_args()
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
dist.init_process_group("gloo")
t = torch.ones(1); dist.all_reduce(t)
print("noise from rank %d" % dist.get_rank())
if dist.get_rank() == 0:
    print(json.dumps({"metric": "stub", "n_gpus": int(os.environ["WORLD_SIZE"]), "ranks": int(t.item()), "gpus_arg": a.gpus,
                      "steps": a.steps, "warmup": a.warmup}))
dist.barrier(); dist.destroy_process_group()
sys.exit(3 if os.environ.get("CCVPE_STUB_FAIL") == "1" else 0)
'''


def _bench(tmp_path, argv, fake_gpus, extra_env=None):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(STUB)
    # the parent only counts devices (from sysfs, without loading the HIP runtime); on this CPU box the count is faked
    wrapper = tmp_path / "run_bench.py"
    wrapper.write_text("import sys, runpy\n"
                       "sys.argv = [%r] + sys.argv[1:]\n"
                       "runpy.run_path(%r, run_name='__main__')\n"
                       "assert 'torch' not in sys.modules, 'the parent must not load torch / HIP'\n"
                       % (os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "bench.py")))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CCVPE_BENCH_CHILD"] = str(stub)
    env["CCVPE_BENCH_FAKE_GPUS"] = str(fake_gpus)          # the parent counts GPUs from sysfs (no HIP): faked here
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(wrapper)] + argv, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)


def test_gpus_2_without_a_launcher_starts_two_ranks_and_relays_the_line(tmp_path):
    res = _bench(tmp_path, ["--gpus", "2", "--steps", "7", "--warmup", "2"], fake_gpus=2)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(out) == 1, "exactly ONE line on stdout (the ranks' other output goes to stderr): %r" % out
    line = json.loads(out[0])
    assert line == {"metric": "stub", "n_gpus": 2, "ranks": 2, "gpus_arg": 2, "steps": 7, "warmup": 2}
    assert "noise from rank" in res.stderr and "torch.distributed.run" in res.stderr


def test_child_failure_is_the_parents_exit_code_and_the_line_still_comes_through(tmp_path):
    res = _bench(tmp_path, ["--gpus", "2", "--steps", "1", "--warmup", "0"], fake_gpus=2, extra_env={"CCVPE_STUB_FAIL": "1"})
    assert res.returncode != 0
    assert json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])["ranks"] == 2


def test_eight_ranks_one_relayed_line(tmp_path):
    """The SCALE driver's largest form, `--gpus 8`, through the launcher path on CPU (gloo): ONE line on stdout, the all-reduce
    counts 8 ranks, and the whole thing is over in bounded time."""
    import time
    t0 = time.time()
    res = _bench(tmp_path, ["--gpus", "8", "--steps", "3", "--warmup", "1"], fake_gpus=8, extra_env={"OMP_NUM_THREADS": "1"})
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    out = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(out) == 1
    assert json.loads(out[0]) == {"metric": "stub", "n_gpus": 8, "ranks": 8, "gpus_arg": 8, "steps": 3, "warmup": 1}
    assert res.stderr.count("noise from rank") == 8
    assert time.time() - t0 < 240


def test_record_glued_to_another_ranks_output_is_cut_out(tmp_path, capfd):
    """torch.distributed.run may deliver rank 0's record and another rank's text on ONE line (seen on this container: the
    8-rank test above read `{...}noise from rank 7`): the parent relays the JSON object alone, the rest goes to stderr."""
    import bench
    glue = tmp_path / "glue_rank.py"
    glue.write_text("import json, sys\n"
                    "sys.stdout.write('prefix ' + json.dumps({'metric': 'stub', 'value': 1.5, 'config': {'a': '{\"x\"}'}}) + 'noise from rank 7\\n')\n")
    rc = bench.launch_ranks(1, ["--gpus", "1"], script=str(glue), timeout=120)
    cap = capfd.readouterr()
    assert rc == 0
    out = [l for l in cap.out.splitlines() if l.strip()]
    assert len(out) == 1 and json.loads(out[0]) == {"metric": "stub", "value": 1.5, "config": {"a": '{"x"}'}}
    assert "noise from rank 7" in cap.err and "prefix" in cap.err


def test_a_launcher_that_dies_at_once_without_a_record_is_started_once_more(tmp_path, capfd):
    import bench
    marker = tmp_path / "attempts"
    die = tmp_path / "die_rank.py"
    die.write_text("import sys\nopen(%r, 'a').write('x')\nsys.exit(5)\n" % str(marker))
    rc = bench.launch_ranks(1, ["--gpus", "1"], script=str(die), timeout=120)
    cap = capfd.readouterr()
    assert rc != 0 and cap.out.strip() == ""
    assert marker.read_text() == "xx" and "one more attempt" in cap.err


def test_timeout_kills_the_whole_rank_group(tmp_path):
    """A hung rank must not outlive the parent's timeout: the launcher runs in its own process group and the group is killed
    (killing torch.distributed.run alone would leave the rank processes holding their GPUs)."""
    import time
    import bench
    marker = tmp_path / "pids"
    hang = tmp_path / "hang_rank.py"
    hang.write_text("import os, time\nopen(%r, 'a').write('%%d\\n' %% os.getpid())\ntime.sleep(600)\n" % str(marker))
    t0 = time.time()
    rc = bench.launch_ranks(2, ["--gpus", "2"], script=str(hang), timeout=20)
    assert rc == 124 and time.time() - t0 < 60
    time.sleep(1.0)
    pids = [int(x) for x in marker.read_text().split()]
    assert len(pids) == 2
    for pid in pids:
        alive = os.path.exists("/proc/%d" % pid) and "Z" not in open("/proc/%d/stat" % pid).read().split()[2]
        assert not alive, "rank process %d survived the timeout" % pid


def test_more_gpus_than_the_node_has_is_refused(tmp_path):
    res = _bench(tmp_path, ["--gpus", "4"], fake_gpus=2)
    assert res.returncode == 2 and "shows 2 GPU" in res.stderr and res.stdout.strip() == ""


def test_world_size_mismatch_is_refused():
    import bench
    import argparse
    import pytest
    with pytest.raises(SystemExit) as e:
        bench.check_world(argparse.Namespace(gpus=8), 1)
    assert e.value.code == 2
    bench.check_world(argparse.Namespace(gpus=2), 2)


def test_side_legs_are_flat_scalars():
    import bench
    ent = {"ms_per_step": 171.5, "value": 373.2, "batch_per_gpu": 64, "dtype": "f32", "loss_after_last_step": 1.0,
           "roofline": {"kernel": "k", "frac": 0.5, "whole_step": {"bound": "mfma", "frac": 0.4}},
           "cpu_baseline": {"value": 0.69, "cores": 32, "sample": "s"}}
    flat = bench.compact("fwd_bwd_vigor_b64", ent)
    assert flat["fwd_bwd_vigor_b64_ms"] == 171.5 and flat["fwd_bwd_vigor_b64_pairs_per_s"] == 373.2
    assert flat["fwd_bwd_vigor_b64_frac"] == 0.4 and flat["fwd_bwd_vigor_b64_cpu_pairs_per_s"] == 0.69
    assert all(not isinstance(v, (dict, list)) for v in flat.values())
    assert set(flat) == {"fwd_bwd_vigor_b64_pairs_per_s", "fwd_bwd_vigor_b64_ms", "fwd_bwd_vigor_b64_frac", "fwd_bwd_vigor_b64_cpu_pairs_per_s"}
    assert set(bench.compact("C2_bf16", {"error": "x"})) == {"c2_bf16_error"}
    c4 = bench.compact("C4_bf16_graph_b256", dict(ent, launch="hipGraph replay", cpu_baseline=None))
    assert set(c4) == {"c4_graph_b256_pairs_per_s", "c4_graph_b256_ms", "c4_graph_b256_frac"}
    # a data-parallel line: 4 descriptive keys + 3 + 1 (fwd+bwd) + 3 x 3 (bf16 legs) + 3 (dp) + 4 (collective) = 24 scalars
    assert 4 + len(flat) + 3 * len(c4) + 3 + len(bench.flat_collective(
        {"ranks_counted_by_allreduce": 8, "allreduce_calls_per_step": 3.0, "bytes_per_step": 1, "backend": "nccl (RCCL)"})) <= 24


def test_per_configuration_work_prices_fov180_below_fov360():
    """C4 (ground 320 x 320) must not be priced with the FoV-360 bytes (profiles/algo_work.json, tools/algo_work.py)."""
    import bench
    g360, mb360 = bench.algo_work("vigor20")
    g180, mb180 = bench.algo_work("vigor_prior180_fov180")
    assert abs(g360 - 56.37) < 0.01 and abs(mb360 - 888.0) < 0.5            # BASELINE.md section 3 stays the yardstick
    assert 0.80 < mb180 / mb360 < 0.92 and 54.0 < g180 < 55.5
    gk, mbk = bench.algo_work("kitti")
    assert abs(gk - 54.45) < 0.01 and abs(mbk - 944.0) < 0.5
    ws360 = bench.whole_step_roof("bf16", 256, "vigor", 44.0, work_key="vigor20")
    ws180 = bench.whole_step_roof("bf16", 256, "vigor", 44.0, work_key="vigor_prior180_fov180")
    assert ws180["frac"] < ws360["frac"] and ws180["bound"] == "hbm"


def test_gpu_count_guard_respects_visibility_masks(monkeypatch):
    """`--gpus N` on a node that SHOWS fewer devices to the ranks (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES) must be
    refused by the parent's guard, not discovered as rank crashes; and the timeout path identifies processes by (pid, start time)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("CCVPE_BENCH_FAKE_GPUS", "8")
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench._count_gpus_without_hip() == 8
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench._count_gpus_without_hip() == 3
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4")
    assert bench._count_gpus_without_hip() == 1
    me = os.getpid()
    assert bench._start_time(me) is not None and bench._start_time(me) == bench._start_time(me)
    assert bench._start_time(2 ** 22 + 12345) is None
