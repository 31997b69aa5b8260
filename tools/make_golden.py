"""Generate tests/golden/*.npz from the REFERENCE (imported from /root/reference).

Runs only in the build container.  Inputs/weights come from ccvpe_amd.synth (bit-exact
integer hash), so the fixtures hold the reference's OUTPUTS only.  Usage:
    python tools/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ref_import import import_reference          # noqa: E402
from ccvpe_amd import synth                       # noqa: E402
import golden_util as G                           # noqa: E402


def save(name, d):
    os.makedirs(G.GOLDEN_DIR, exist_ok=True)
    path = os.path.join(G.GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **{k: np.ascontiguousarray(v) for k, v in d.items()})
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def main():
    torch.set_num_threads(8)
    ref_models, ref_losses = import_reference()
    sds = {}
    with torch.no_grad():
        # ---- full forwards --------------------------------------------------------------
        only_train = "--only-train" in sys.argv
        only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only-case=")]
        for name, c in ({} if only_train else G.FORWARD_CASES).items():
            if only and name not in only:
                continue
            key = (c["kind"], c["wseed"])
            if key not in sds:
                sds[key] = synth.synthetic_state_dict(*key)
            sd = sds[key]
            if c["kind"] == "kitti":
                net = ref_models.CVM_KITTI("cpu")
            elif c["kind"] == "oxford":
                net = ref_models.CVM_OxfordRobotCar("cpu")
            elif c["ori_noise"] is None:
                net = ref_models.CVM_VIGOR("cpu", c["circular"])
            else:
                net = ref_models.CVM_VIGOR_ori_prior("cpu", c["ori_noise"], c["circular"])
            net.load_state_dict(sd, strict=True)
            net.eval()
            grd, sat = synth.synthetic_pair(c["batch"], c["grd"], c["pseed"])
            out = net(grd, sat)
            d = G.summarize_forward(out)
            # encoder / descriptor intermediates straight from the reference sub-modules
            gf = net.grd_efficientnet.extract_features(grd)
            sv, ms = net.sat_efficientnet.extract_features_multiscale(sat)
            d["grd_feature_c8"] = gf[:, ::8].numpy()
            d["sat_feature_c8"] = sv[:, ::8].numpy()
            for i in (0, 2, 4, 10, 15):
                st = max(1, ms[i].shape[-1] // 16)
                d["sat_block%d_s" % i] = ms[i][:, :, ::st, ::st].numpy()
            for l in range(1, 7):
                d["grd_desc%d" % l] = getattr(net, "grd_feature_to_descriptor%d" % l)(gf).numpy()
            save("fwd_" + name, d)

        if only:
            return
        # ---- train-mode forward (batch-stat BN, injected drop_connect draws) ----------------------------
        import efficientnet_pytorch.model as effmodel
        train_only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--train-case=")]
        for tname, c in G.TRAIN_CASES.items():
            if train_only and tname not in train_only:
                continue
            if (c["kind"], c["wseed"]) not in sds:
                sds[(c["kind"], c["wseed"])] = synth.synthetic_state_dict(c["kind"], c["wseed"])
            sd = sds[(c["kind"], c["wseed"])]
            masks, _, skip = G.train_drop_masks(c["batch"])
            order = [("grd_efficientnet", i) for i in skip] + [("sat_efficientnet", i) for i in skip]
            calls = []

            def injected_drop_connect(inputs, p, training):
                key = order[len(calls)]
                calls.append(key)
                return inputs / (1 - p) * masks[key].view(-1, 1, 1, 1)
            real_dc = effmodel.drop_connect
            effmodel.drop_connect = injected_drop_connect
            net = ref_models.CVM_KITTI("cpu") if c["kind"] == "kitti" else ref_models.CVM_VIGOR("cpu", c["circular"])
            net.load_state_dict(sd, strict=True)
            net.train()
            grd, sat = synth.synthetic_pair(c["batch"], c["grd"], c["pseed"])
            with torch.enable_grad():
                out = net(grd, sat)
                effmodel.drop_connect = real_dc
                assert len(calls) == len(order)
                # ---- gradients of the shared deterministic loss through the REFERENCE (autograd on the CPU) --------
                G.train_loss(out).backward()
            save("grad_%s_trainmode" % tname, G.summarize_grads([(n, p.grad) for n, p in net.named_parameters()]))
            out = [t.detach() for t in out]
            d = G.summarize_forward(out)
            after = net.state_dict()
            for k in G.RUNNING_STAT_SAMPLES:
                d["rm:" + k] = after[k + ".running_mean"].numpy()
                d["rv:" + k] = after[k + ".running_var"].numpy()
            save("fwd_%s_trainmode" % tname, d)
        if only_train:
            return

        # ---- single MBConv blocks / stem on small inputs -----------------------------------
        sd = sds[("vigor", 0)]
        for circ, pfx in ((True, "grd_efficientnet"), (False, "sat_efficientnet")):
            net = ref_models.CVM_VIGOR("cpu", circ)
            net.load_state_dict(sd, strict=True)
            net.eval()
            eff = getattr(net, pfx)
            d = {}
            x = synth.normal((2, 3, 32, 48), 4242)
            d["stem"] = eff._swish(eff._bn0(eff._conv_stem(x))).numpy()
            for i, (k, s, e, cin, cout) in enumerate(synth.B0_BLOCKS):
                if not circ and i not in G.ZERO_PAD_BLOCKS:
                    continue
                xin = synth.normal((2, cin) + G.BLOCK_HW, 5000 + i)
                d["block%d" % i] = eff._blocks[i](xin).numpy()
            xin = synth.normal((2, 320, 5, 6), 6000)
            d["head"] = eff._swish(eff._bn1(eff._conv_head(xin))).numpy()
            save("effnet_modules_" + ("circ" if circ else "zero"), d)

        # ---- losses ---------------------------------------------------------------------------
        d = {}
        for n_cols in (1280, 20480):
            sc = synth.uniform((3, n_cols), 8000 + n_cols, -1.0, 1.0)
            lab = synth.uniform((3, n_cols), 8100 + n_cols) ** 6
            d["infonce_%d" % n_cols] = ref_losses.infoNCELoss(sc, lab).numpy()
        lg = synth.normal((3, 262144), 8200, 2.0)
        lab = synth.uniform((3, 262144), 8201) ** 20
        lab = lab / lab.sum(1, keepdim=True)
        d["ce"] = ref_losses.cross_entropy_loss(lg, lab).numpy()
        ori = torch.nn.functional.normalize(synth.normal((3, 2, 512, 512), 8300), dim=1)
        gto = torch.nn.functional.normalize(synth.normal((3, 2, 512, 512), 8301), dim=1)
        d["ori"] = ref_losses.orientation_loss(ori, gto, lab.reshape(3, 1, 512, 512)).numpy()
        save("losses", d)


if __name__ == "__main__":
    main()
