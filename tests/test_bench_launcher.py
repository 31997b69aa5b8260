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
    # the parent only counts devices; on this CPU box the count is faked through a sitecustomize-free hook: a tiny wrapper
    # that patches torch.cuda.device_count before running bench.main()
    wrapper = tmp_path / "run_bench.py"
    wrapper.write_text("import sys, runpy, torch\n"
                       "torch.cuda.device_count = lambda: %d\n"
                       "sys.argv = [%r] + sys.argv[1:]\n"
                       "runpy.run_path(%r, run_name='__main__')\n" % (fake_gpus, os.path.join(ROOT, "bench.py"),
                                                                      os.path.join(ROOT, "bench.py")))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CCVPE_BENCH_CHILD"] = str(stub)
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
    assert set(bench.compact("C2_bf16", {"error": "x"})) == {"c2_bf16_error"}
    assert "c4_graph_b256_ms" in bench.compact("C4_bf16_graph_b256", dict(ent, launch="hipGraph replay"))
