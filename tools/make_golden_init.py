"""tests/golden/init_stats.npz: per-tensor statistics of FRESHLY CONSTRUCTED reference models (CVM_VIGOR, CVM_KITTI;
pretrained download replaced by a no-op, i.e. exactly what `EfficientNet.from_name` + torch's default layer initialisation
give) — the fixture tests/test_init.py compares ccvpe_amd's construction-time initialisation against.
Runs only in the build container (imports /root/reference):  python tools/make_golden_init.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ref_import import import_reference          # noqa: E402
import golden_util as G                           # noqa: E402

ref_models, _ = import_reference()
out = {}
for kind, ctor in (("vigor", lambda: ref_models.CVM_VIGOR("cpu", True)), ("kitti", lambda: ref_models.CVM_KITTI("cpu"))):
    net = ctor()
    sd = net.state_dict()
    names = list(sd.keys())
    out[kind + ":names"] = np.array(names)
    out[kind + ":numel"] = np.array([sd[k].numel() for k in names], dtype=np.int64)
    out[kind + ":mean"] = np.array([float(sd[k].double().mean()) for k in names])
    out[kind + ":std"] = np.array([float(sd[k].double().std()) if sd[k].numel() > 1 else 0.0 for k in names])
    out[kind + ":absmax"] = np.array([float(sd[k].double().abs().max()) for k in names])
path = os.path.join(G.GOLDEN_DIR, "init_stats.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
