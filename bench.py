#!/usr/bin/env python3
"""bench.py — CCVPE dense cross-view matching path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches it under
torch.distributed.run (one rank per GPU, RCCL).  A "step" of the headline is ONE forward pass of the hot path
over one batch of synthetic image pairs resident in HBM.

Workload at N=1 = BASELINE.json configs[1] ("C1"): CVM_VIGOR_ori_prior(ori_noise=0) (N_rot=1 in the localisation
branch, 20 in the orientation branch), batch 64 per GPU, fp32, ground 3x320x640 + aerial 3x512x512, synthetic inputs
and seeded random-init weights (no network).  Inference shards by sample: N>1 runs N replicas with NO data-path
collective ("weak" scaling); the only collectives are the timing barrier and the max-over-ranks of the elapsed time.

One JSON line on rank 0 with, besides the contract fields:
  roofline     — for the dominant kernel (largest share of the step among the HIP-event-bracketed dense launches):
                 achieved = algorithmic FLOPs (fp32: MFMA-bound) or bytes (bf16: HBM-bound) of its launches in the timed
                 region / their HIP-event durations; peak from MI355X_MICROARCH.md; traffic = PMC HBM bytes per launch from
                 the committed profiles/ pass (`traffic_source` names it), else null; `whole_step` prices the entire step.
  cpu_baseline — the CPU oracle (oracle/ccvpe_oracle.py, kind "port") on the host cores, bounded sample, rank 0 / N=1.
  config       — besides the workload description, the side measurements the default command also runs, as FLAT scalar keys
                 (the driver's record keeps the scalars of `config`; nested objects are dropped): `<leg>_ms`,
                 `<leg>_pairs_per_s`, `<leg>_frac` (whole step against its governing roof), `<leg>_kernel[_frac]`, ... for
                 leg = fwd_bwd_vigor_b64 (BASELINE metric's "(fwd+bwd) VIGOR bs=64": the full training step, with
                 `_cpu_pairs_per_s`), c2_bf16 (configs[2]), c1_bf16 (C1 model in bf16 storage), c4_graph_b256 (configs[4]);
                 for N>1 additionally train_dp_kitti_b64 (configs[3]: the data-parallel training step with the RCCL
                 gradient all-reduce) and `collective_*` (backend, ranks as RCCL counts them, all-reduce calls and bytes
                 per step).  A failure of the data-parallel leg prints the line and then EXITS NON-ZERO.
The per-kernel tables of every leg go to stderr (one JSON object per leg, prefix "[bench kernels]"), not into the line.
`--train` makes the training step the headline (`--model kitti` = C3).

`--gpus N` without a launcher (WORLD_SIZE unset) and N > 1: this process — before anything touches a GPU — starts the N ranks
itself (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
as a child process), relays rank 0's line and exits with the child's code.  Under a launcher WORLD_SIZE must equal --gpus.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MATRIX_PEAK_TFLOPS = 2500.0     # same table: ~2.5 PF dense
HBM_PEAK_GBS = 8000.0
# algorithmic forward work per image pair (2 x MACs of every conv / deconv / linear; N_rot = 20 / 16)
GFLOP_PER_PAIR = {"vigor": 56.37, "kitti": 54.45}
MB_PER_PAIR_FP32 = {"vigor": 888.0, "kitti": 944.0}
# per-CONFIGURATION work (FoV 180 halves the ground encoder, ...): profiles/algo_work.json, written by tools/algo_work.py with
# BASELINE.md §3's rule (hooked B = 1 oracle forward).  Its byte totals are 1.5 % above BASELINE.md's for the two models priced
# there (901.2 vs 888 MB, 959.6 vs 944), so they are scaled to BASELINE.md's: the published constants stay the yardstick and a
# configuration is priced by its RATIO to the FoV-360 model of its family.
WORK_KEY = {"prior0": "vigor_prior0", "vigor20": "vigor20", "prior180_fov180": "vigor_prior180_fov180", "kitti": "kitti",
            "oxford": "oxford"}
_ALGO = None


def algo_work(work_key):
    """(GFLOP per pair, fp32 activation MB per pair) of a benched configuration, or None."""
    global _ALGO
    if _ALGO is None:
        try:
            _ALGO = json.load(open(os.path.join(ROOT, "profiles", "algo_work.json")))
        except Exception:
            _ALGO = {}
    d = _ALGO.get(work_key)
    if not d:
        return None
    fam = "kitti" if work_key == "kitti" else "vigor"
    ref = _ALGO.get("kitti" if fam == "kitti" else "vigor20")
    mb = d["mb_per_pair_fp32"] * (MB_PER_PAIR_FP32[fam] / ref["mb_per_pair_fp32"]) if ref else d["mb_per_pair_fp32"]
    return d["gflop_per_pair"], mb


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="image pairs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the forward from a captured hipGraph")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="fp32 = BASELINE C1 (default, the headline); bf16 = the C2/C4 storage path")
    ap.add_argument("--model", choices=["prior0", "vigor20", "prior180_fov180", "kitti", "oxford"], default="prior0",
                    help="prior0 = C1 (default); vigor20 = C2 (N_rot=20); prior180_fov180 = C4; kitti = C3 forward")
    ap.add_argument("--per-layer", action="store_true", help="print a per-launch-shape table to stderr")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (config.<leg> summaries)")
    ap.add_argument("--legs", default="c2,c1bf16,c4,train,dp",
                    help="side measurements of the default command: c2, c1bf16, c4, train (fwd+bwd VIGOR), dp (data-parallel "
                         "KITTI step; runs only under a process group)")
    ap.add_argument("--train", action="store_true",
                    help="time full training steps (forward + losses + backward + gradient all-reduce + Adam) instead "
                         "of the eval forward: BASELINE config C3 with --model kitti")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket the dense launches with HIP events in the timed region")
    ap.add_argument("--exact-infonce", action="store_true",
                    help="--train with N>1: all-reduce infoNCE's numerator / label mass so that the loss equals the "
                         "single-process big-batch loss (losses.py:18) instead of the mean of per-rank losses")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------
# CPU baselines (oracle = test infrastructure; used here only as the timed CPU leg)
# ------------------------------------------------------------------------------------------------------
def _cpu_threads():
    """The per-op work of a small-batch forward does not feed more than a few dozen cores (256 threads ran 50x SLOWER
    than 32 in a first measurement); `cores` reports the threads actually used."""
    return min(os.cpu_count() or 1, 32)


def cpu_baseline(sd, batch=8, reps=3):
    """Oracle forward on the host cores: bounded sample (~10-30 s)."""
    import torch
    from ccvpe_amd import synth
    from oracle import ccvpe_oracle as O
    cores = _cpu_threads()
    torch.set_num_threads(cores)
    grd, sat = synth.synthetic_pair(batch, "vigor", 1234)
    times = []
    with torch.no_grad():
        O.forward(sd, grd, sat, "vigor", True, ori_noise=0)          # warm-up
        for _ in range(reps):
            t0 = time.perf_counter()
            O.forward(sd, grd, sat, "vigor", True, ori_noise=0)
            times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return dict(value=batch / med, unit="img-pairs/s", cores=cores, kind="port",
                sample="oracle forward, CVM_VIGOR_ori_prior(0), B=%d fp32, 1 warm-up + median of %d" % (batch, reps))


def cpu_baseline_train(sd, kind="vigor", batch=2, reps=2):
    """Oracle training step on the host cores (train-mode forward, the reference's loss mix, autograd backward):
    train_VIGOR.py:112-150 without the optimizer update.  Bounded sample (~15-30 s)."""
    import torch
    from ccvpe_amd import synth
    from oracle import ccvpe_oracle as O
    cores = _cpu_threads()
    torch.set_num_threads(cores)
    n_rot = synth.MODEL_SPECS[kind]["n_rot"]
    grd, sat = synth.synthetic_pair(batch, kind, 1234)
    u = synth.uniform((batch, 3), 99)
    gt, gt_flat, gt_ori, labels = O.train_targets(((u[:, :2] - 0.5) * 384.0).tolist(), (u[:, 2] * 359.99).tolist(), n_rot)
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone())
              for k, v in sd.items()}

    def step():
        for v in params.values():
            if v.grad is not None:
                v.grad = None
        out = O.forward(params, grd, sat, kind, True, None, train_stats={})
        nce = 0.0
        for lvl in range(6):
            nce = nce + O.infonce_loss(out[3 + lvl].flatten(1), labels[lvl].flatten(1))
        loss = O.cross_entropy_loss(out[0], gt_flat) + 1e4 * nce / 6 + 1e1 * O.orientation_loss(out[2], gt_ori, gt)
        loss.backward()

    step()
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return dict(value=batch / med, unit="img-pairs/s", cores=cores, kind="port",
                sample="oracle train forward + losses + autograd backward, %s, B=%d fp32, median of %d"
                       % (kind, batch, reps))


# ------------------------------------------------------------------------------------------------------
# roofline from the HIP-event launch records
# ------------------------------------------------------------------------------------------------------
def _traffic_table(workload):
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.isfile(tpath):
        return {}
    try:
        return json.load(open(tpath)).get(workload, {})
    except Exception:
        return {}


def traffic_key(name, table):
    """The entry of a pmc_traffic.json workload table a recorder name resolves to: the name itself, else the kernel's family
    name (template arguments dropped: the counter pass keys some families without them), else None."""
    if name in table:
        return name
    base = name.split("<", 1)[0]
    return base if base in table else None


def _traffic(name, workload):
    """PMC HBM bytes per launch of kernel `name` from the committed counter pass of `workload` ("f32" forward, "bf16"
    forward, "bf16_c2" or "train"), with its provenance."""
    d = _traffic_table(workload)
    if not d:
        return None, None
    meta = d.get("#meta", {})
    key = traffic_key(name, d)
    src = "profiles/pmc_traffic.json[%s][%s] @%s (static rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes, not re-measured here)" \
          % (workload, key, meta.get("commit", "unknown"))
    return (d.get(key) if key else None), src


def pmc_step_bytes(workload, ms_step):
    """What the COUNTERS say a whole step of this workload moves (sum over every kernel family of the committed pass: bytes per
    launch x launches per step, written by tools/pmc_traffic.py as #meta.gb_per_step) and the HBM utilisation that is at this
    run's step time — next to the rule-based `frac`, which prices the step with ALGORITHMIC bytes."""
    meta = _traffic_table(workload).get("#meta", {})
    gb = meta.get("gb_per_step")
    if not gb or not ms_step:
        return {}
    return {"pmc_gb_per_step": round(gb, 2), "hbm_util": round(gb / ms_step / (HBM_PEAK_GBS / 1e3), 4),
            "pmc_source": "profiles/pmc_traffic.json[%s] @%s" % (workload, meta.get("commit", "unknown"))}


def whole_step_roof(precision, batch, kind, ms_step, train=False, work_key=None, pmc_workload=None):
    """The whole step against the governing roof (forward GFLOP / activation MB per pair; fwd+bwd =
    3 x forward FLOPs).  Needs only the wall time of a step: also available for hipGraph replays.  `work_key` prices the step
    with ITS configuration's work (profiles/algo_work.json) instead of the family's FoV-360 constants."""
    aw = algo_work(work_key) if work_key else None
    gf = aw[0] if aw else GFLOP_PER_PAIR.get(kind)
    mb32 = aw[1] if aw else MB_PER_PAIR_FP32.get(kind)
    if gf is None or not ms_step:
        return None
    mult = 3.0 if train else 1.0
    step_tf = mult * gf * batch / ms_step          # GFLOP / ms = TFLOP/s
    ws = {"algorithmic_tflop_per_step": round(mult * gf * batch / 1e3, 3), "achieved_tflops": round(step_tf, 2)}
    if precision == "bf16":
        step_gbs = mb32 / 2 * batch / ms_step        # MB / ms = GB/s
        ws.update(bound="hbm", algorithmic_gb_per_step=round(mb32 / 2 * batch / 1e3, 2),
                  achieved_GBps=round(step_gbs, 1), frac=round(step_gbs / HBM_PEAK_GBS, 4),
                  mfma_bf16_frac=round(step_tf / BF16_MATRIX_PEAK_TFLOPS, 4))
    else:
        ws.update(bound="mfma", frac=round(step_tf / FP32_MATRIX_PEAK_TFLOPS, 4))
    if pmc_workload:
        ws.update(pmc_step_bytes(pmc_workload, ms_step))
    return ws


def roofline_from(summ, steps, precision, batch, kind, ms_step, train=False, work_key=None, pmc_workload=None):
    """summ: ops.LaunchRecorder.summary() (None / empty for a hipGraph replay: events cannot be recorded inside one — the
    roofline then prices the whole step only).  Dominant kernel = the largest total time among the recorded dense launches.
    Returns (roofline dict, per-kernel table or None)."""
    if pmc_workload is None:
        pmc_workload = "train" if train else ("f32" if precision == "fp32" else None)
    ws = whole_step_roof(precision, batch, kind, ms_step, train, work_key, pmc_workload)
    if ws is not None and work_key:
        ws["work"] = work_key
    if not summ:
        if ws is None:
            return None, None
        if ws["bound"] == "hbm":
            roof = {"bound": "hbm", "kernel": "(whole step: graph replay has no per-launch events)", "achieved": ws["achieved_GBps"],
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ws["frac"], "traffic": None}
        else:
            roof = {"bound": "mfma", "kernel": "(whole step: graph replay has no per-launch events)", "achieved": ws["achieved_tflops"],
                    "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ws["frac"], "traffic": None}
        roof["whole_step"] = ws
        return roof, None
    tot_ms = sum(d["ms"] for d in summ.values())
    name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
    tfl = d["flops"] / (d["ms"] * 1e-3) / 1e12
    gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
    traffic, tsrc = _traffic(name, "train" if train else ("f32" if precision == "fp32" else "bf16"))
    kernel_is_f32 = "f32" in name or name.startswith("conv_wgrad")           # the bf16 path's fp32 tail runs fp32 kernels
    if precision == "bf16" and not kernel_is_f32:
        # bf16 storage: the whole forward is HBM-governed (18 k pairs/s HBM vs 44 k MFMA); the dominant
        # kernel is priced against BOTH roofs, `bound` names the one its arithmetic intensity puts it under
        ai = d["flops"] / max(d["bytes"], 1.0)
        bound = "mfma" if ai > BF16_MATRIX_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS else "hbm"
        head = {"bound": bound, "kernel": name}
        if bound == "hbm":
            head.update(achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4))
        else:
            head.update(achieved=round(tfl, 1), peak=BF16_MATRIX_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(tfl / BF16_MATRIX_PEAK_TFLOPS, 4))
        head["vs_both_roofs"] = {"hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "mfma_bf16_frac": round(tfl / BF16_MATRIX_PEAK_TFLOPS, 4),
                                 "algo_GBps": round(gbs, 1), "tflops": round(tfl, 1)}
    else:
        head = {"bound": "mfma", "kernel": name, "achieved": round(tfl, 2), "peak": FP32_MATRIX_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tfl / FP32_MATRIX_PEAK_TFLOPS, 4)}
        if precision == "bf16":
            head["note"] = "fp32 kernel of the bf16 path's fp32 tail (set_precision fp32_tail_levels): priced against the fp32 matrix peak"
    head["traffic"] = traffic
    head["traffic_source"] = tsrc if traffic is not None else None
    roof = dict(head)
    roof.update(launches_per_step=d["calls"] // steps, avg_launch_ms=round(d["ms"] / d["calls"], 4),
                algorithmic_gflop_per_launch=round(d["flops"] / d["calls"] / 1e9, 3),
                algorithmic_mb_per_launch=round(d["bytes"] / d["calls"] / 1e6, 2),
                share_of_recorded_time=round(d["ms"] / tot_ms, 3))
    # the three largest families by recorded time, each against its own roof (a "dominant" kernel alone is 8-10 % of a step)
    top3 = []
    for nm, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:3]:
        t_tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
        t_gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9
        is_f32 = "f32" in nm or nm.startswith("conv_wgrad") or precision == "fp32"
        if not is_f32:
            t_ai = v["flops"] / max(v["bytes"], 1.0)
            t_bound = "mfma" if t_ai > BF16_MATRIX_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS else "hbm"
            t_frac = t_tf / BF16_MATRIX_PEAK_TFLOPS if t_bound == "mfma" else t_gb / HBM_PEAK_GBS
        else:
            t_bound, t_frac = "mfma", t_tf / FP32_MATRIX_PEAK_TFLOPS
        top3.append({"kernel": nm, "share_of_recorded_time": round(v["ms"] / tot_ms, 3), "bound": t_bound, "frac": round(t_frac, 4),
                     "ms_per_step": round(v["ms"] / steps, 3)})
    roof["top3"] = top3
    if ws is not None:
        roof["whole_step"] = ws
    table = {k: {"ms_per_step": round(v["ms"] / steps, 3),
                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                 "algo_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                 "launches_per_step": v["calls"] // steps}
             for k, v in sorted(summ.items())}
    return roof, table


def emit_kernel_table(tag, table):
    """Per-kernel table of one leg -> stderr (kept out of the JSON line: the driver's 2 000-character tail must show the
    line's head, and its record keeps `config` / `roofline` only)."""
    if table:
        print("[bench kernels] " + json.dumps({tag: table}), file=sys.stderr)
        sys.stderr.flush()


LEG_KEY = {"fwd_bwd_vigor_b64": "fwd_bwd_vigor_b64", "C2_bf16": "c2_bf16", "C1_bf16": "c1_bf16",
           "C4_bf16_graph_b256": "c4_graph_b256", "train_dp_kitti_b64": "train_dp_kitti_b64"}


def compact(tag, entry):
    """Driver-visible summary of a side measurement: FLAT scalar keys `<leg>_<field>` directly under `config`.  The driver's
    record keeps the first 24 scalars of `config` and drops nested objects, so a leg gets exactly three — `_pairs_per_s`,
    `_ms`, `_frac` (whole step against its governing roof, priced with the leg's own algorithmic work) — plus the CPU figure
    of the fwd+bwd leg; everything else about the leg goes to stderr (`emit_leg_details`)."""
    k = LEG_KEY.get(tag, tag.lower())
    if "error" in entry:
        return {k + "_error": entry["error"][:300]}
    out = {k + "_pairs_per_s": entry["value"], k + "_ms": entry["ms_per_step"]}
    ws = (entry.get("roofline") or {}).get("whole_step") or {}
    out[k + "_frac"] = ws.get("frac")
    if ws.get("hbm_util") is not None:
        out[k + "_hbm_util"] = ws.get("hbm_util")          # what the COUNTERS say (profiles/pmc_traffic.json), beside the byte-rule frac
    cb = entry.get("cpu_baseline")
    if cb:
        out[k + "_cpu_pairs_per_s"] = round(cb["value"], 3)
    return out


def emit_leg_details(tag, entry):
    """Everything `compact` leaves out (batch, dtype, bound, dominant kernel and its fraction, CPU sample, loss, peak HBM,
    launch mode) -> stderr as one JSON object, prefix "[bench leg]"."""
    d = {k: v for k, v in entry.items() if k not in ("roofline",)}
    r = entry.get("roofline") or {}
    d["roofline"] = {k: v for k, v in r.items()}
    print("[bench leg] " + json.dumps({LEG_KEY.get(tag, tag.lower()): d}), file=sys.stderr)
    sys.stderr.flush()


def flat_collective(coll):
    """Four scalars (a data-parallel line has 4 + 4 legs x 3 + 1 + 4 = 21 scalar config keys)."""
    return {"collective_ranks": coll["ranks_counted_by_allreduce"], "collective_calls_per_step": coll["allreduce_calls_per_step"],
            "collective_bytes_per_step": coll["bytes_per_step"], "collective_backend": coll["backend"]}


def per_layer_table(rec, steps):
    agg = {}
    for name, tag, flops, nbytes, e0, e1 in rec.items:
        d = agg.setdefault((name, tag), [0, 0.0, 0.0, 0.0])
        d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += flops; d[3] += nbytes
    print("%-30s %-34s %5s %9s %8s %8s" % ("kernel", "shape", "n/st", "ms/step", "TFLOP/s", "GB/s"), file=sys.stderr)
    for (name, tag), d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-30s %-34s %5d %9.3f %8.1f %8.0f" % (name, tag, d[0] // steps, d[1] / steps,
              d[2] / d[1] / 1e9, d[3] / d[1] / 1e6), file=sys.stderr)


# ------------------------------------------------------------------------------------------------------
# measurements
# ------------------------------------------------------------------------------------------------------
def forward_measure(fwd, grd, sat, dev, steps, warmup, record):
    """Times `steps` forwards; returns (max-over-ranks seconds, recorder or None)."""
    import torch
    from ccvpe_amd import harness, ops
    rec = ops.LaunchRecorder() if record else None
    state = {"n": 0}

    def step():
        # the recorder is switched on after the warm-up steps so that it brackets exactly the launches of the timed region
        if state["n"] == warmup:
            ops.set_recorder(rec)
        state["n"] += 1
        fwd(grd, sat)

    try:
        elapsed = harness.timed_steps(step, steps, warmup, sync_fn=torch.cuda.synchronize, device=dev)
    finally:
        ops.set_recorder(None)
    return elapsed, rec


def train_measure(net, grd, sat, dev, batch, steps, warmup, rank, n_rot, record=False, exact_infonce=False):
    """Times `steps` training steps as train_VIGOR.py:112-150 / train_KITTI.py:120-151 run them: ground truth, forward
    (train mode), the three losses on all levels, backward, data-parallel gradient averaging (RCCL all-reduce), Adam.
    Returns dict(elapsed = max-over-ranks seconds, loss, peak_gib, rec, allreduce_calls_per_step, allreduce_bytes_per_step)."""
    import torch
    from ccvpe_amd import harness, losses, ops, optim, synth, targets

    net.train()
    # ground truth as datasets.py builds it (Gaussian sigma 4 px at a seeded offset, orientation bins, (cos, sin) map) and
    # train_VIGOR.py:120-128 pools it — generated on the device from 3 scalars per sample (ccvpe_train_targets_f32)
    u = synth.uniform((batch, 3), 99 + rank)
    center = ((u[:, :2] - 0.5) * 384.0).to(dev)
    angle = (u[:, 2] * 359.99).to(dev)
    opt = optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))                 # train_VIGOR.py:104, one launch per step
    reducer = harness.GradientAllReducer(net.parameters()).attach(net, optimizer=opt)     # all-reduce overlapped with the backward
    nce_fn = losses.infoNCELoss_global if exact_infonce else losses.infoNCELoss
    last = {}
    rec = ops.LaunchRecorder() if record else None
    state = {"n": 0, "calls0": 0}

    def step():
        if state["n"] == warmup:
            ops.set_recorder(rec)
            state["calls0"] = reducer.allreduce_calls
        state["n"] += 1
        opt.zero_grad(set_to_none=True)
        gt, gt_flat, gt_ori, labels = targets.train_targets(center, angle, n_rot)
        out = net(grd, sat)
        nce = 0.0
        for lvl in range(6):                                                        # train_VIGOR.py:137-146
            nce = nce + nce_fn(torch.flatten(out[3 + lvl], start_dim=1), torch.flatten(labels[lvl], start_dim=1))
        loss = losses.cross_entropy_loss(out[0], gt_flat) + 1e4 * nce / 6 + 1e1 * losses.orientation_loss(out[2], gt_ori, gt)
        loss.backward()
        reducer()
        opt.step()
        last["loss"] = loss.detach()

    torch.cuda.reset_peak_memory_stats(dev)
    try:
        elapsed = harness.timed_steps(step, steps, warmup, sync_fn=torch.cuda.synchronize, device=dev)
    finally:
        ops.set_recorder(None)
    loss = float(last["loss"])
    if loss != loss or loss in (float("inf"), float("-inf")):
        raise RuntimeError("training step produced a non-finite loss (%r)" % loss)
    return dict(elapsed=elapsed, loss=loss, peak_gib=torch.cuda.max_memory_allocated(dev) / 2 ** 30, rec=rec,
                allreduce_calls_per_step=(reducer.allreduce_calls - state["calls0"]) / float(steps),
                allreduce_bytes_per_step=reducer.bytes_reduced_last_step)


def train_entry(net, kind, grd, sat, dev, batch, steps, warmup, rank, world, n_rot, record, exact_infonce=False, tag="train"):
    """One training measurement as a JSON-able dict (value = whole-job pairs/s) + its roofline."""
    m = train_measure(net, grd, sat, dev, batch, steps, warmup, rank, n_rot, record, exact_infonce)
    ms = 1e3 * m["elapsed"] / steps
    out = {"workload": "%s training step (device-side ground truth, train-mode forward, loss = CE + 1e4 * mean of 6 infoNCE + "
                       "10 * orientation, backward, %sAdam lr 1e-4), batch %d per GPU, fp32"
                       % (type(net).__name__, "RCCL gradient all-reduce in 3 groups overlapped with the backward, "
                          if world > 1 else "", batch),
           "value": round(batch * world * steps / m["elapsed"], 2), "unit": "img-pairs/s", "n_gpus": world,
           "batch_per_gpu": batch, "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup, "dtype": "f32",
           "loss_after_last_step": round(m["loss"], 5), "peak_hbm_gib": round(m["peak_gib"], 2)}
    roof, table = roofline_from(m["rec"].summary() if m["rec"] is not None else None, steps, "fp32", batch, kind, ms, train=True,
                                work_key="kitti" if kind == "kitti" else "vigor20")
    out["roofline"] = roof
    if rank == 0:
        emit_kernel_table(tag, table)
    return out, m


def harness_active():
    from ccvpe_amd import harness
    return harness.GradientAllReducer.active()


def collective_info(dev, m, world):
    """What the collective layer actually did in the timed region of a data-parallel training leg (N > 1)."""
    import torch
    import torch.distributed as dist
    ones = torch.ones((1,), device=dev)
    dist.all_reduce(ones)                                   # every rank adds 1: the world size as RCCL itself counts it
    return {"backend": "%s (RCCL)" % dist.get_backend(), "world_size_env": world, "ranks_counted_by_allreduce": int(ones.item()),
            "allreduce_calls_per_step": m["allreduce_calls_per_step"], "bytes_per_step": int(m["allreduce_bytes_per_step"]),
            "op": "ncclAvg in place on the flat gradient arena, 3 groups, issued from inside the backward"}


# ------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: start the ranks as a child process (this process never creates a GPU context)
# ------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, script=None, timeout=None, _attempt=0):
    """Runs `bench.py argv` as n ranks under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1), relays the
    children's stderr and rank 0's stdout, returns the launcher's exit code.  `script` (or CCVPE_BENCH_CHILD, tests only)
    replaces bench.py as the program the ranks run.  A launcher that dies within 20 s without a record (the rendezvous port
    taken between _free_port() and the bind: seen once in ~30 back-to-back runs of the CPU tests) is started once more."""
    import subprocess
    t_start = time.time()
    script = script or os.environ.get("CCVPE_BENCH_CHILD") or os.path.abspath(__file__)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script] + list(argv)
    print("bench.py: --gpus %d without a launcher: starting %s" % (n, " ".join(cmd)), file=sys.stderr)
    sys.stderr.flush()
    # The launcher gets its own session (= process group): on a timeout or an interrupt the WHOLE group is killed — killing only
    # torch.distributed.run would leave the GPU rank processes (its grandchildren) alive and holding the GPUs.  stdout goes to a
    # temporary file, not a pipe: nothing is buffered in this process while the ranks run.
    import signal
    import tempfile
    with tempfile.TemporaryFile(mode="w+") as out:
        proc = subprocess.Popen(cmd, env=env, stdout=out, text=True, start_new_session=True)
        try:
            rc = proc.wait(timeout=timeout)
        except (subprocess.TimeoutExpired, KeyboardInterrupt) as ex:
            # torch.distributed.run starts every rank in a session of its OWN, so the launcher's process group does not contain
            # them: collect the descendants first, ask the launcher to stop (its SIGTERM handler stops its workers), then kill
            # whatever is still alive — launcher group and every rank by exact pid
            ranks = [(pid, _start_time(pid)) for pid in _descendants(proc.pid)]
            try:
                os.killpg(proc.pid, signal.SIGTERM)
            except ProcessLookupError:
                pass
            try:
                proc.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
            # a pid may have been recycled during the 10 s: kill only what is still THE SAME process (same start time in
            # /proc/<pid>/stat), and a process group only when it is the rank's own session (pgid == pid)
            victims = [(proc.pid, _start_time(proc.pid))] + ranks
            for pid, born in victims:
                if born is None or _start_time(pid) != born:
                    continue
                try:
                    if os.getpgid(pid) == pid:
                        os.killpg(pid, signal.SIGKILL)
                    os.kill(pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
            proc.wait()
            print("bench.py: %s: killed the launcher and its %d rank process(es)" % (type(ex).__name__, len(ranks)), file=sys.stderr)
            return 124
        out.seek(0)
        stdout = out.read()
    # The ranks' stdout reaches this file through torch.distributed.run, which may write a rank's line in pieces next to another
    # rank's (seen here: `{...}noise from rank 7` on one line, the newline of rank 0's record arriving later): the record is
    # therefore CUT OUT of its line as the JSON object that parses, and whatever surrounds it goes to stderr with the other noise.
    lines = []
    dec = json.JSONDecoder()
    for l in stdout.splitlines():
        at = l.find('{"')
        rec = None
        while at >= 0 and rec is None:
            try:
                obj, end = dec.raw_decode(l[at:])
                if isinstance(obj, dict) and "metric" in obj:
                    rec = (at, at + end)
            except ValueError:
                pass
            if rec is None:
                at = l.find('{"', at + 1)
        if rec is None:
            print(l, file=sys.stderr)
            continue
        lines.append(l[rec[0]:rec[1]])
        rest = (l[:rec[0]] + " " + l[rec[1]:]).strip()
        if rest:
            print(rest, file=sys.stderr)
    if lines:
        print(lines[-1])
        sys.stdout.flush()
    if rc == 0 and not lines:
        print("bench.py: the ranks exited 0 without printing a line", file=sys.stderr)
        return 1
    if rc != 0 and not lines and _attempt == 0 and time.time() - t_start < 20.0:
        print("bench.py: the launcher exited %d after %.1f s without a record: one more attempt on a new port" % (rc, time.time() - t_start),
              file=sys.stderr)
        return launch_ranks(n, argv, script=script, timeout=timeout, _attempt=1)
    return rc


def _start_time(pid):
    """Start time (clock ticks since boot, field 22 of /proc/<pid>/stat) — identifies a process across pid reuse; None if gone."""
    try:
        with open("/proc/%d/stat" % pid) as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except (OSError, ValueError, IndexError):
        return None


def _descendants(root):
    """pids of every live descendant of `root` (from /proc: ppid chains), children before grandchildren."""
    kids = {}
    for d in os.listdir("/proc"):
        if d.isdigit():
            try:
                with open("/proc/%s/stat" % d) as f:
                    ppid = int(f.read().rsplit(")", 1)[1].split()[1])
            except (OSError, ValueError, IndexError):
                continue
            kids.setdefault(ppid, []).append(int(d))
    out, todo = [], [root]
    while todo:
        for k in kids.get(todo.pop(0), []):
            out.append(k)
            todo.append(k)
    return out


def _count_gpus_without_hip():
    """GPUs of this node WITHOUT loading the HIP runtime in this (parent) process: the KFD topology lists one node per agent,
    GPUs are the nodes with SIMDs.  None if the topology is not readable (then the ranks themselves will fail loudly)."""
    if os.environ.get("CCVPE_BENCH_FAKE_GPUS"):                 # tests only (CPU boxes have no KFD topology)
        n = int(os.environ["CCVPE_BENCH_FAKE_GPUS"])
    else:
        top = "/sys/class/kfd/kfd/topology/nodes"
        try:
            n = 0
            for d in os.listdir(top):
                with open(os.path.join(top, d, "properties")) as f:
                    for line in f:
                        k, _, v = line.partition(" ")
                        if k == "simd_count" and int(v) > 0:
                            n += 1
        except (OSError, ValueError):
            return None
    # a visibility mask narrows what the ranks will see: the guard must not pass on devices they cannot open
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:            # (an EMPTY mask hides every device: the ranks then fail loudly by themselves; the CPU tests use it)
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def check_world(args, world):
    """The contract's --gpus N must be the number of ranks that are really running."""
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a %d-GPU number from %d rank(s)"
              % (args.gpus, world, args.gpus, world), file=sys.stderr)
        sys.exit(2)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and args.gpus > 1:
        # nothing in this process touches the GPU (not even the HIP runtime is loaded: the GPUs are counted from sysfs)
        have = _count_gpus_without_hip()
        if have is not None and have < args.gpus:
            print("bench.py: --gpus %d but this node shows %d GPU(s)" % (args.gpus, have), file=sys.stderr)
            sys.exit(2)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    check_world(args, world)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under a launcher (RANK set) the process group is created even for one rank: `torchrun --nproc-per-node 1` then
    # exercises the exact collective code path of the multi-GPU runs (tests/test_bench_torchrun_gpu.py)
    use_pg = world > 1 or "RANK" in os.environ
    if use_pg:
        dist.init_process_group("nccl", device_id=dev)      # RCCL

    from ccvpe_amd import models, ops, synth, _lib
    _lib.load()                                             # fails loudly if the HIP library is missing
    torch.manual_seed(1234 + rank)                          # drop_connect draws of the training legs: the same in every run

    kind = args.model if args.model in ("kitti", "oxford") else "vigor"
    sd = synth.synthetic_state_dict(kind, 0)                # identical on every rank
    if args.model == "prior0":
        net, gshape = models.CVM_VIGOR_ori_prior(dev, 0, True), "vigor"
    elif args.model == "vigor20":
        net, gshape = models.CVM_VIGOR(dev, True), "vigor"
    elif args.model == "prior180_fov180":
        net, gshape = models.CVM_VIGOR_ori_prior(dev, 180, False), "vigor_fov180"
    elif args.model == "oxford":
        net, gshape = models.CVM_OxfordRobotCar(dev), "oxford"
    else:
        net, gshape = models.CVM_KITTI(dev), "kitti"
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval().set_precision(args.precision)
    grd, sat = synth.synthetic_pair(args.batch, gshape, 1234 + rank, device=dev)      # generated in HBM (same bits as on the CPU): resident before timing
    record = not args.no_kernel_events
    failed = []                                             # legs whose failure must fail the run (after the line is printed)

    def finish(code=0):
        if use_pg:
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:                               # noqa: BLE001 — already failing
                pass
        if code:
            sys.exit(code)

    # ---------------------------------------------------------------------------------------------------
    if args.train:
        if args.precision != "fp32" or args.graph:
            raise SystemExit("--train is fp32, eager only")
        n_rot = synth.MODEL_SPECS[kind]["n_rot"]
        entry, m = train_entry(net, kind, grd, sat, dev, args.batch, args.steps, args.warmup, rank, world, n_rot, record,
                               args.exact_infonce)
        coll = collective_info(dev, m, world) if use_pg else None
        if rank == 0:
            line = {"metric": "train image-pairs/sec (fwd+bwd)", "value": entry["value"], "unit": "img-pairs/s", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": entry["ms_per_step"], "higher_is_better": True,
                    "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": ("C3: " if kind == "kitti" else "") + entry["workload"],
                               "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                               "parallelism": "dp%d (RCCL all-reduce of gradients in 3 groups, overlapped with the backward; %s)"
                                              % (world, "exact big-batch infoNCE (2-scalar all-reduce per level)"
                                                 if args.exact_infonce else "per-rank loss means, as torch DDP"),
                               "loss_after_last_step": entry["loss_after_last_step"], "peak_hbm_gib": entry["peak_hbm_gib"],
                               "weights": "seeded random init, reference state_dict layout"},
                    "roofline": entry.get("roofline"), "cpu_baseline": None}
            if coll is not None:
                line["config"].update(flat_collective(coll))
            if m["rec"] is not None and args.per_layer:
                per_layer_table(m["rec"], args.steps)
            if world == 1 and not args.no_cpu_baseline and kind in ("vigor", "kitti"):
                line["cpu_baseline"] = cpu_baseline_train(sd, kind)
            print(json.dumps(line))
            sys.stdout.flush()
        finish()
        return

    # ---------------------------------------------------------------------------------------------------
    if args.graph:
        from ccvpe_amd.graph import GraphedForward
        fwd = GraphedForward(net, grd, sat)
        record = False                          # events cannot be recorded inside a replayed graph
    else:
        fwd = net
    elapsed, rec = forward_measure(fwd, grd, sat, dev, args.steps, args.warmup, record)

    line = None
    if rank == 0:
        pairs = args.batch * world * args.steps
        value = pairs / elapsed
        ms_step = 1e3 * elapsed / args.steps
        line = {
            "metric": "image-pairs/sec", "value": round(value, 2), "unit": "img-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": {"prior0": "C1: CVM_VIGOR_ori_prior(ori_noise=0) eval forward, "
                                              "grd 3x320x640 + sat 3x512x512, N_rot=1 loc / 20 ori",
                                    "vigor20": "C2: CVM_VIGOR eval forward, N_rot=20, grd 3x320x640 + sat 3x512x512",
                                    "prior180_fov180": "C4: CVM_VIGOR_ori_prior(180, circular_padding=False) eval "
                                                       "forward, FoV 180: grd 3x320x320 + sat 3x512x512",
                                    "kitti": "C3 (forward only): CVM_KITTI eval forward, grd 3x256x1024 + sat "
                                             "3x512x512",
                                    "oxford": "CVM_OxfordRobotCar eval forward (SURVEY 8(f)-3), grd 3x154x231 + sat "
                                              "3x512x512, N_rot=20"}[args.model],
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": "replicas x%d (no data-path collective)" % world},
        }
        # (the driver's record keeps 24 scalars of `config`: launch mode and weights join the workload text)
        line["config"]["workload"] += "; %s; seeded random-init weights in the reference state_dict layout" % (
            "hipGraph replay" if args.graph else "eager launches (one C-ABI call per kernel)")
        if args.model == "oxford" and args.batch == 1 and world == 1:
            # the reference's only published rate for this path family: "14 FPS" per-frame pose estimation with the Oxford
            # RobotCar variant, hardware not stated (/root/reference/README.md:21)
            line["vs_baseline"] = round(value / 14.0, 2)
            line["config"]["baseline"] = "14 FPS per frame (reference README, hardware not stated)"
        pmc_wl = None
        if args.precision == "bf16" and args.batch == {"prior0": 64, "vigor20": 32}.get(args.model):
            pmc_wl = {"prior0": "bf16", "vigor20": "bf16_c2"}[args.model]     # the configurations the counter passes were taken on
        elif args.precision == "fp32" and not (args.model == "prior0" and args.batch == 64):
            pmc_wl = ""                                                       # another fp32 configuration: no counter pass for it
        line["roofline"], table = roofline_from(rec.summary() if rec is not None else None, args.steps, args.precision,
                                                args.batch, kind, ms_step, work_key=WORK_KEY.get(args.model), pmc_workload=pmc_wl)
        emit_kernel_table("headline", table)
        if rec is not None and args.per_layer:
            per_layer_table(rec, args.steps)
        if world == 1 and not args.no_cpu_baseline and args.model == "prior0":
            line["cpu_baseline"] = cpu_baseline(sd)
        else:
            line["cpu_baseline"] = None

    default_run = (args.model == "prior0" and args.precision == "fp32" and not args.graph and not args.no_extra)
    if default_run:
        legs = {}
        want = set(args.legs.split(","))
        del fwd
        # (4) BASELINE.json's metric string names "(fwd+bwd) VIGOR bs=64": the full training step of CVM_VIGOR (N_rot = 20)
        #     at batch 64 per GPU: ground truth, train-mode forward, losses, backward, (all-reduce,) Adam
        try:
            del net                                 # (first among the side legs: right after the fp32 headline, before the
            torch.cuda.empty_cache()                # bf16 legs and the B = 256 graph have churned the allocator)
            if "train" not in want:
                raise KeyError("skipped")
            net3 = models.CVM_VIGOR(dev, True)
            net3.load_state_dict(sd, strict=True)
            net3 = net3.to(dev)
            # 3 warm-up steps: the 2nd captures the re-pack hipGraph, and the caching allocator's pools of the three streams of
            # the training step stop growing by the 3rd
            ent, _ = train_entry(net3, "vigor", grd, sat, dev, args.batch, 5, 3, rank, world, 20, record,
                                 tag="fwd_bwd_vigor_b64")
            if rank == 0:
                if world == 1 and not args.no_cpu_baseline:
                    ent["cpu_baseline"] = cpu_baseline_train(sd, "vigor")
                legs["fwd_bwd_vigor_b64"] = ent
            del net3
        except KeyError:
            pass
        except Exception as ex:
            if rank == 0:
                legs["fwd_bwd_vigor_b64"] = {"error": repr(ex)}
        # (1) BASELINE configs[2] "C2": CVM_VIGOR, N_rot = 20, batch 32, bf16 storage; (2) the C1 model in bf16 at batch 64
        #     (the north-star's ">= 10 000 pairs/s forward at batch 64" is a bf16 goal: fp32 MFMA ceiling is 2.8 k);
        # (3) configs[4] "C4": ori_prior(180), FoV 180 ground input, bf16, hipGraph replay at batch 256
        for tag, ctor, b2, gsh, seed, graph, what in (
                ("C2_bf16", lambda: models.CVM_VIGOR(dev, True), 32, "vigor", 4321, False,
                 "C2: CVM_VIGOR eval forward, N_rot=20, batch 32, bf16 storage (fp32 accumulate)"),
                ("C1_bf16", lambda: models.CVM_VIGOR_ori_prior(dev, 0, True), args.batch, "vigor", 1234, False,
                 "C1 model CVM_VIGOR_ori_prior(0) eval forward, batch %d, bf16 storage (fp32 accumulate)" % args.batch),
                ("C4_bf16_graph_b256", lambda: models.CVM_VIGOR_ori_prior(dev, 180, False), 256, "vigor_fov180", 777, True,
                 "C4: CVM_VIGOR_ori_prior(180, circular_padding=False), FoV 180, batch 256, bf16 storage, hipGraph replay")):
            if {"C2_bf16": "c2", "C1_bf16": "c1bf16", "C4_bf16_graph_b256": "c4"}[tag] not in want:
                continue
            try:
                net2 = ctor()
                net2.load_state_dict(sd, strict=True)
                net2 = net2.to(dev).eval().set_precision("bf16")
                g2, s2 = synth.synthetic_pair(b2, gsh, seed + rank, device=dev)
                f2 = net2
                if graph:
                    from ccvpe_amd.graph import GraphedForward
                    f2 = GraphedForward(net2, g2, s2)
                # timed pass WITHOUT per-launch events (two events per launch are ~0.3-0.5 ms of a 7-12 ms bf16 step: ~300
                # launches), then a short pass with them for the per-kernel table / dominant kernel of this leg
                ns2 = 10                            # (10 timed steps after 4 warm-ups: with 5 + 2 the caching allocator was still
                e2, _ = forward_measure(f2, g2, s2, dev, ns2, 4, False)     # settling after the training leg: C2 6.2 vs 5.9 ms standalone)
                r2 = None
                if record and not graph:
                    _, r2 = forward_measure(f2, g2, s2, dev, 3, 1, True)
                if rank == 0:
                    roof, table = roofline_from(r2.summary() if r2 is not None else None, 3, "bf16", b2, "vigor", 1e3 * e2 / ns2,
                                                work_key={"C2_bf16": "vigor20", "C1_bf16": "vigor_prior0",
                                                          "C4_bf16_graph_b256": "vigor_prior180_fov180"}[tag],
                                                pmc_workload={"C2_bf16": "bf16_c2", "C1_bf16": "bf16"}.get(tag))
                    emit_kernel_table(tag, table)
                    legs[tag] = {"workload": what, "value": round(b2 * world * ns2 / e2, 2), "n_gpus": world, "batch_per_gpu": b2,
                                 "ms_per_step": round(1e3 * e2 / ns2, 3), "steps": ns2, "dtype": "bf16", "roofline": roof,
                                 "launch": "hipGraph replay" if graph else "eager"}
                del net2, g2, s2, r2, f2
                torch.cuda.empty_cache()
            except Exception as ex:      # the headline must not depend on a side measurement; the error is driver-visible
                if rank == 0:
                    legs[tag] = {"error": repr(ex)}
        # (5) under a process group (N > 1): BASELINE configs[3] "C3" — CVM_KITTI data-parallel training step, batch 64 per
        #     GPU, gradients averaged by RCCL all-reduce over xGMI.  This is the ONLY leg with a data-path collective: if it
        #     fails the line is still printed (with the error) and the process exits non-zero.
        if use_pg and "dp" in want:
            coll = None
            try:
                torch.cuda.empty_cache()
                if os.environ.get("CCVPE_BENCH_FAIL_DP") == "1":        # tests: the failure path of this leg
                    raise RuntimeError("injected failure (CCVPE_BENCH_FAIL_DP=1)")
                sdk = synth.synthetic_state_dict("kitti", 0)
                net4 = models.CVM_KITTI(dev)
                net4.load_state_dict(sdk, strict=True)
                net4 = net4.to(dev)
                g4, s4 = synth.synthetic_pair(args.batch, "kitti", 1234 + rank, device=dev)
                dp_steps, dp_warm = (int(v) for v in os.environ.get("CCVPE_BENCH_DP_STEPS", "5,3").split(","))    # (tests shorten it)
                ent, m4 = train_entry(net4, "kitti", g4, s4, dev, args.batch, dp_steps, dp_warm, rank, world, 16, record,
                                      tag="train_dp_kitti_b64")
                coll = collective_info(dev, m4, world)
                if coll["ranks_counted_by_allreduce"] != world:
                    raise RuntimeError("all-reduce counted %d ranks, WORLD_SIZE is %d" % (coll["ranks_counted_by_allreduce"], world))
                if harness_active() and coll["allreduce_calls_per_step"] < 1:
                    raise RuntimeError("no gradient all-reduce ran in the timed region")
                if rank == 0:
                    legs["train_dp_kitti_b64"] = ent
            except Exception as ex:
                failed.append("train_dp_kitti_b64: %r" % (ex,))
                if rank == 0:
                    legs["train_dp_kitti_b64"] = {"error": repr(ex)}
            if rank == 0 and coll is not None:
                line["config"].update(flat_collective(coll))
        if rank == 0:
            # order = the order the driver's 24 config scalars are spent in: BASELINE's literal metric first, then the bf16 legs,
            # then (N > 1) the data-parallel step and what the collective layer did
            coll_flat = {k: line["config"].pop(k) for k in list(line["config"]) if k.startswith("collective_")}
            for tag in ("fwd_bwd_vigor_b64", "C1_bf16", "C2_bf16", "C4_bf16_graph_b256", "train_dp_kitti_b64"):
                if tag in legs:
                    line["config"].update(compact(tag, legs[tag]))
                    emit_leg_details(tag, legs[tag])
            line["config"].update(coll_flat)
            # the counters' view of each leg goes LAST (the driver keeps the first 24 scalars: at N > 1 the legs + collective_* are
            # exactly 24, and these must not push them out)
            for k in [k for k in line["config"] if k.endswith("_hbm_util")]:
                line["config"][k] = line["config"].pop(k)
            # the literal BASELINE metric's dominant kernel, flat, beside the headline's roofline (nested objects are dropped)
            fb = (legs.get("fwd_bwd_vigor_b64") or {}).get("roofline") or {}
            if fb.get("kernel"):
                line["roofline"].update({"fwd_bwd_kernel": fb["kernel"], "fwd_bwd_kernel_achieved": fb.get("achieved"),
                                         "fwd_bwd_kernel_frac": fb.get("frac"), "fwd_bwd_kernel_avg_launch_ms": fb.get("avg_launch_ms"),
                                         "fwd_bwd_kernel_traffic": fb.get("traffic")})
    if rank == 0:
        ws = (line.get("roofline") or {}).get("whole_step") or {}
        if ws.get("hbm_util") is not None:          # flat copy (the driver's record drops nested objects)
            line["roofline"].update({"whole_step_frac": ws.get("frac"), "whole_step_pmc_gb": ws.get("pmc_gb_per_step"),
                                     "whole_step_hbm_util": ws.get("hbm_util")})
        print(json.dumps(line))
        sys.stdout.flush()
    if failed:
        print("bench.py: FAILED legs: " + "; ".join(failed), file=sys.stderr)
    finish(1 if failed else 0)


if __name__ == "__main__":
    main()
