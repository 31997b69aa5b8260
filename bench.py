#!/usr/bin/env python3
"""bench.py — CCVPE dense cross-view matching forward on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches it under
torch.distributed.run (one rank per GPU, RCCL).  A "step" is ONE forward pass of the hot path
over one batch of synthetic image pairs resident in HBM.

Workload at N=1 = BASELINE.json configs[1] ("C1"): CVM_VIGOR_ori_prior(ori_noise=0)
(N_rot=1 in the localisation branch, 20 in the orientation branch), batch 64 per GPU, fp32,
ground 3x320x640 + aerial 3x512x512, synthetic inputs and seeded random-init weights (no network).
Inference shards by sample: N>1 runs N replicas with NO data-path collective ("weak" scaling);
the only collectives are the timing barrier and the max-over-ranks of the elapsed time.

One JSON line on rank 0 with, besides the contract fields:
  roofline     — for the dominant kernel (the fp32-MFMA implicit-GEMM instantiation with the
                 largest share of the step): achieved = algorithmic FLOPs of its launches in the
                 timed region / their HIP-event durations; peak = 157.3 TF fp32 matrix
                 (MI355X_MICROARCH.md); traffic = PMC HBM bytes per launch if a profiles/ pass
                 recorded them, else null.
  cpu_baseline — the CPU oracle (oracle/ccvpe_oracle.py, kind "port") on the host cores, bounded
                 sample, rank 0 / N=1 only.
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
HBM_PEAK_GBS = 8000.0
GFLOP_PER_PAIR = {"vigor": 56.37}    # BASELINE.md §3 (N_rot=20); reported in config for reference


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
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the short bf16 C2 side measurement attached as `extra` to the default N=1 line")
    ap.add_argument("--train", action="store_true",
                    help="time full training steps (forward + losses + backward + gradient all-reduce + Adam) instead "
                         "of the eval forward: BASELINE config C3 with --model kitti")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket igemm launches with HIP events in the timed region")
    return ap.parse_args()


def cpu_baseline(sd, batch=8, reps=3):
    """Oracle forward on the host cores: bounded sample (~10-30 s).  The thread count is capped:
    the per-op work of a B=8 forward does not feed more than a few dozen cores (256 threads ran
    50x SLOWER than 32 in a first measurement); `cores` reports the threads actually used."""
    import torch
    from ccvpe_amd import synth
    from oracle import ccvpe_oracle as O
    cores = min(os.cpu_count() or 1, 32)
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


def train_measure(net, grd, sat, dev, batch, steps, warmup, rank, n_rot):
    """Times `steps` training steps as train_VIGOR.py:193-229 / train_KITTI.py run them: ground truth, forward (train mode),
    the three losses on all levels, backward, data-parallel gradient averaging (RCCL all-reduce), Adam.
    Returns (max-over-ranks seconds, last loss, peak HBM GiB)."""
    import torch
    from ccvpe_amd import harness, losses, optim, synth, targets

    class _A(object):
        pass
    args = _A()
    args.batch, args.steps, args.warmup = batch, steps, warmup
    net.train()
    # ground truth as datasets.py builds it (Gaussian sigma 4 px at a seeded offset, orientation bins, (cos, sin) map) and
    # train_VIGOR.py:120-128 pools it — generated on the device from 3 scalars per sample (ccvpe_train_targets_f32)
    u = synth.uniform((args.batch, 3), 99 + rank)
    center = ((u[:, :2] - 0.5) * 384.0).to(dev)
    angle = (u[:, 2] * 359.99).to(dev)
    opt = optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))                 # train_VIGOR.py:104, one launch per step
    reducer = harness.GradientAllReducer(net.parameters()).attach(net)     # all-reduce overlapped with the backward
    last = {}

    def step():
        opt.zero_grad(set_to_none=True)
        gt, gt_flat, gt_ori, labels = targets.train_targets(center, angle, n_rot)
        out = net(grd, sat)
        nce = 0.0
        for lvl in range(6):                                                        # train_VIGOR.py:137-146
            nce = nce + losses.infoNCELoss(torch.flatten(out[3 + lvl], start_dim=1), torch.flatten(labels[lvl], start_dim=1))
        loss = losses.cross_entropy_loss(out[0], gt_flat) + 1e4 * nce / 6 + 1e1 * losses.orientation_loss(out[2], gt_ori, gt)
        loss.backward()
        reducer()
        opt.step()
        last["loss"] = loss.detach()

    torch.cuda.reset_peak_memory_stats(dev)
    elapsed = harness.timed_steps(step, args.steps, args.warmup, sync_fn=torch.cuda.synchronize, device=dev)
    return elapsed, float(last["loss"]), torch.cuda.max_memory_allocated(dev) / 2 ** 30


def train_main(args, net, grd, sat, dev, world, rank, n_rot):
    import torch
    elapsed, loss, peak = train_measure(net, grd, sat, dev, args.batch, args.steps, args.warmup, rank, n_rot)
    if rank == 0:
        line = {
            "metric": "train image-pairs/sec", "value": round(args.batch * world * args.steps / elapsed, 2),
            "unit": "img-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3: %s training step (device-side ground truth, train-mode forward, loss = CE + 1e4 * mean of 6 "
                                   "infoNCE + 10 * orientation, backward, gradient all-reduce, Adam lr 1e-4)" % type(net).__name__,
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": "dp%d (RCCL all-reduce of gradients in 3 groups, overlapped with the backward)" % world,
                       "loss_after_last_step": round(loss, 5), "peak_hbm_gib": round(peak, 2)},
            "roofline": None, "cpu_baseline": None,
        }
        print(json.dumps(line))
        sys.stdout.flush()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)      # RCCL

    from ccvpe_amd import models, ops, synth, _lib
    _lib.load()                                             # fails loudly if the HIP library is missing

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
    grd, sat = synth.synthetic_pair(args.batch, gshape, 1234 + rank)
    grd, sat = grd.to(dev), sat.to(dev)                     # inputs resident in HBM before timing
    if args.train:
        if args.precision != "fp32" or args.graph:
            raise SystemExit("--train is fp32, eager only")
        train_main(args, net, grd, sat, dev, world, rank, synth.MODEL_SPECS[kind]["n_rot"])
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from ccvpe_amd import harness
    if args.graph:
        from ccvpe_amd.graph import GraphedForward
        fwd = GraphedForward(net, grd, sat)
        args.no_kernel_events = True            # events cannot be recorded inside a replayed graph
    else:
        fwd = net
    rec = None if args.no_kernel_events else ops.LaunchRecorder()
    state = {"n": 0}

    def step():
        # the recorder is switched on after the warm-up steps so that it brackets exactly the
        # igemm launches of the timed region
        if state["n"] == args.warmup:
            ops.set_recorder(rec)
        state["n"] += 1
        fwd(grd, sat)

    elapsed = harness.timed_steps(step, args.steps, args.warmup, sync_fn=torch.cuda.synchronize, device=dev)
    ops.set_recorder(None)

    if rank == 0:
        pairs = args.batch * world * args.steps
        value = pairs / elapsed
        line = {
            "metric": "image-pairs/sec", "value": round(value, 2), "unit": "img-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
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
                       "parallelism": "replicas x%d (no data-path collective)" % world,
                       "launch": "hipGraph replay" if args.graph else "eager (one C-ABI call per kernel)",
                       "weights": "seeded random init (ccvpe_amd.synth), reference state_dict layout"},
        }
        if args.model == "oxford" and args.batch == 1 and world == 1:
            # the reference's only published rate for this path family: "14 FPS" per-frame pose estimation with the Oxford
            # RobotCar variant, hardware not stated (/root/reference/README.md:21)
            line["vs_baseline"] = round(value / 14.0, 2)
            line["config"]["baseline"] = "14 FPS per frame (reference README, hardware not stated)"
        roof = None
        if rec is not None:
            summ = rec.summary()
            tot_ms = sum(d["ms"] for d in summ.values())
            name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.isfile(tpath):
                try:
                    traffic = json.load(open(tpath)).get(name)
                except Exception:
                    traffic = None
            if args.precision == "bf16":
                # bf16 storage: every kernel is HBM-bound (MFMA is 16x faster than fp32) -> price the
                # dominant kernel against HBM with its ALGORITHMIC bytes (input + output + weights once)
                ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                roof_head = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic}
            else:
                roof_head = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2),
                             "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(achieved / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": traffic}
            roof = {**roof_head,
                    "launches_per_step": d["calls"] // args.steps,
                    "avg_launch_ms": round(d["ms"] / d["calls"], 4),
                    "algorithmic_gflop_per_launch": round(d["flops"] / d["calls"] / 1e9, 3),
                    "share_of_igemm_time": round(d["ms"] / tot_ms, 3),
                    "all_igemm": {k: {"ms_per_step": round(v["ms"] / args.steps, 3),
                                      "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                      "algo_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                      "launches_per_step": v["calls"] // args.steps}
                                  for k, v in sorted(summ.items())}}
        line["roofline"] = roof
        if rec is not None and args.per_layer:
            agg = {}
            for name, tag, flops, nbytes, e0, e1 in rec.items:
                d = agg.setdefault((name, tag), [0, 0.0, 0.0, 0.0])
                d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += flops; d[3] += nbytes
            print("%-28s %-28s %5s %9s %8s %8s" % ("kernel", "shape", "n/st", "ms/step", "TFLOP/s", "GB/s"), file=sys.stderr)
            for (name, tag), d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print("%-28s %-28s %5d %9.3f %8.1f %8.0f" % (name, tag, d[0] // args.steps, d[1] / args.steps,
                      d[2] / d[1] / 1e9, d[3] / d[1] / 1e6), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline and args.model == "prior0":
            line["cpu_baseline"] = cpu_baseline(sd)
        else:
            line["cpu_baseline"] = None
        default_run = (world == 1 and args.model == "prior0" and args.precision == "fp32" and not args.graph
                       and not args.no_extra)
        if default_run:
            # side measurement, NOT the headline: BASELINE config C2 (CVM_VIGOR, N_rot = 20, batch 32,
            # bf16 storage path).  Same harness, 5 timed steps.
            try:
                del fwd
                net2 = models.CVM_VIGOR(dev, True)
                net2.load_state_dict(sd, strict=True)
                net2 = net2.to(dev).eval().set_precision("bf16")
                g2, s2 = synth.synthetic_pair(32, "vigor", 4321)
                g2, s2 = g2.to(dev), s2.to(dev)
                e2 = harness.timed_steps(lambda: net2(g2, s2), 5, 2, sync_fn=torch.cuda.synchronize, device=dev)
                line["extra"] = {"C2_bf16": {"workload": "CVM_VIGOR eval forward, N_rot=20, batch 32, bf16 storage "
                                                         "(fp32 accumulate), grd 3x320x640 + sat 3x512x512",
                                             "value": round(32 * 5 / e2, 2), "unit": "img-pairs/s",
                                             "ms_per_step": round(1e3 * e2 / 5, 3), "steps": 5, "dtype": "bf16",
                                             "parity": "tests/test_bf16_gpu.py: logits within 5e-2 of range vs fp32 "
                                                       "oracle, scores within 2e-2"}}
            except Exception as ex:      # the headline must not depend on the side measurement
                line["extra"] = {"C2_bf16": {"error": repr(ex)}}
            # second side measurement: BASELINE.json's metric string also names "(fwd+bwd) VIGOR bs=64" — the full training
            # step of CVM_VIGOR (N_rot = 20) at batch 64: ground truth, train-mode forward, losses, backward, Adam
            try:
                del net2, g2, s2
                del net
                torch.cuda.empty_cache()
                net3 = models.CVM_VIGOR(dev, True)
                net3.load_state_dict(sd, strict=True)
                net3 = net3.to(dev)
                e3, loss3, peak3 = train_measure(net3, grd, sat, dev, args.batch, 3, 1, rank, 20)
                line["extra"]["train_fwd_bwd_vigor_b64"] = {
                    "workload": "CVM_VIGOR training step (device-side ground truth, train-mode forward, CE + 1e4 * mean infoNCE "
                                "+ 10 * orientation, backward, Adam), batch %d, fp32" % args.batch,
                    "value": round(args.batch * 3 / e3, 2), "unit": "img-pairs/s", "ms_per_step": round(1e3 * e3 / 3, 3),
                    "steps": 3, "dtype": "f32", "peak_hbm_gib": round(peak3, 2),
                    "parity": "tests/test_train_backward_gpu.py: gradients vs the reference's autograd golden (520 tensors)"}
            except Exception as ex:
                line["extra"]["train_fwd_bwd_vigor_b64"] = {"error": repr(ex)}
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
