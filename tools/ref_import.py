"""Import the upstream CCVPE reference (read-only, /root/reference) in THIS container.

Test/fixture infrastructure only: used by tools/make_golden.py and the `-m "not gpu"`
oracle-vs-reference tests when /root/reference exists. Never imported by the product
(ccvpe_amd/) and never available on the GPU box.

Two shims:
  1. models.py:1-13 imports torchvision / IPython / tensorboard, none of which any model
     class uses -> register empty stub modules.
  2. EfficientNet.from_pretrained downloads ImageNet weights (utils.py:747) -> replace
     efficientnet_pytorch.model.load_pretrained_weights by a no-op.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("CCVPE_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "models.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns (models_module, losses_module) of the reference."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
        tv.models = _stub("torchvision.models")
        tv.utils = _stub("torchvision.utils", make_grid=None, save_image=None)
    if "IPython" not in sys.modules:
        ip = _stub("IPython")
        ip.display = _stub("IPython.display", Image=None)
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        _stub("torch.utils.tensorboard", SummaryWriter=None)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import efficientnet_pytorch.model as effmodel
    effmodel.load_pretrained_weights = lambda *a, **k: None
    import models as ref_models
    import losses as ref_losses
    return ref_models, ref_losses


def import_reference_datasets():
    """The reference's datasets.py (needs one more stub: torchvision.transforms.functional, imported but unused by the
    VIGOR / KITTI dataset classes)."""
    import_reference()
    tv = sys.modules["torchvision"]
    if not hasattr(tv.transforms, "functional"):
        tv.transforms.functional = _stub("torchvision.transforms.functional")
    import datasets as ref_datasets
    return ref_datasets
