"""Import the reference implementation (/root/reference) in the BUILD CONTAINER to pin the oracle.

TEST INFRASTRUCTURE ONLY; never imported by the product, by `-m gpu` tests, by smoke() or by bench.py
(/root/reference does not exist on the GPU box).  Used by tests/golden/make_golden.py and by the
`-m "not gpu"` tests that compare oracle/bcos_oracle.py with the live reference when it is present.

Why shims are needed: `import bcos` runs bcos/__init__.py:5-20, which eagerly imports
torchvision, pytorch_lightning and torchmetrics -- third-party packages that are neither part of the reference
tree nor installed in this image.  The hot-path files themselves are pure torch, so
  1. `bcos`, `bcos.models`, `CLIP`, `CLIP.clip` are pre-registered as bare namespace modules whose __path__
     points into /root/reference, which makes `import bcos.modules`, `import bcosify`, ... execute the
     reference's own files without the package __init__ files;
  2. the two torchvision symbols those files touch get stand-ins with the published semantics:
     torchvision.transforms.Normalize (clone, sub_ mean, div_ std) and torchvision.models.ResNet (+ blocks),
     the latter being the plain-torch topology restated in b-cosification_amd/bcos/models/_tv_resnet.py.
No reference source is copied; nothing here is used at run time by the product.
"""
import importlib
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("BCOS_REFERENCE_ROOT", "/root/reference")
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_TV_RESNET_FILE = os.path.join(_REPO, "b-cosification_amd", "bcos", "models", "_tv_resnet.py")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "bcos", "modules"))


def _bare(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _load_tv_resnet():
    spec = importlib.util.spec_from_file_location("_tv_resnet_standin", _TV_RESNET_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _install_torchvision_standin():
    import torch

    if "torchvision" in sys.modules and not getattr(sys.modules["torchvision"], "_bcos_standin", False):
        return  # a real torchvision is importable: use it
    tv = types.ModuleType("torchvision")
    tv._bcos_standin = True
    tv.__path__ = []
    transforms = types.ModuleType("torchvision.transforms")
    transforms.__path__ = []
    functional = types.ModuleType("torchvision.transforms.functional")

    def normalize(tensor, mean, std, inplace=False):
        if not inplace:
            tensor = tensor.clone()
        mean = torch.as_tensor(mean, dtype=tensor.dtype, device=tensor.device)
        std = torch.as_tensor(std, dtype=tensor.dtype, device=tensor.device)
        return tensor.sub_(mean.view(-1, 1, 1)).div_(std.view(-1, 1, 1))

    class Normalize(torch.nn.Module):
        def __init__(self, mean, std, inplace=False):
            super().__init__()
            self.mean, self.std, self.inplace = mean, std, inplace

        def forward(self, tensor):
            return normalize(tensor, self.mean, self.std, self.inplace)

    functional.normalize = normalize
    transforms.Normalize = Normalize
    transforms.functional = functional
    models = types.ModuleType("torchvision.models")
    models.__path__ = []
    tvr = _load_tv_resnet()
    resnet = types.ModuleType("torchvision.models.resnet")
    for n in ("ResNet", "BasicBlock", "Bottleneck"):
        setattr(resnet, n, getattr(tvr, n))
    models.ResNet = tvr.ResNet
    models.resnet = resnet

    class DenseNet(torch.nn.Module):   # only subclassed at import time by standard_models.py; never built here
        pass

    models.DenseNet = DenseNet
    tv.transforms, tv.models = transforms, models
    sys.modules.update({"torchvision": tv, "torchvision.transforms": transforms,
                        "torchvision.transforms.functional": functional, "torchvision.models": models,
                        "torchvision.models.resnet": resnet})


_ready = False


def setup():
    """Make `import bcos.modules`, `import bcosify`, `import bcosify_vit`, `import CLIP.clip.model` resolve to the
    reference.  Must run in a process that has NOT imported the product's `bcos` package."""
    global _ready
    if _ready:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if "bcos" in sys.modules and REFERENCE_ROOT not in str(getattr(sys.modules["bcos"], "__path__", "")):
        raise RuntimeError("the product's `bcos` package is already imported in this process")
    try:
        import torchvision  # noqa: F401
    except Exception:
        _install_torchvision_standin()
    _bare("bcos", os.path.join(REFERENCE_ROOT, "bcos"))
    _bare("bcos.models", os.path.join(REFERENCE_ROOT, "bcos", "models"))
    _bare("bcos.data", os.path.join(REFERENCE_ROOT, "bcos", "data"))
    _bare("CLIP", os.path.join(REFERENCE_ROOT, "CLIP"))
    _bare("CLIP.clip", os.path.join(REFERENCE_ROOT, "CLIP", "clip"))
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _ready = True


def modules():
    """-> namespace with the reference classes used by the golden generator."""
    setup()
    ns = types.SimpleNamespace()
    ns.bcos_modules = importlib.import_module("bcos.modules")
    ns.common = importlib.import_module("bcos.common")
    ns.bcosifyconv2d = importlib.import_module("bcos.modules.bcosifyconv2d")
    ns.bcosifylinear = importlib.import_module("bcos.modules.bcosifylinear")
    ns.bcosify = importlib.import_module("bcosify")
    ns.standard_models = importlib.import_module("bcos.models.standard_models")
    ns.tv_resnet = sys.modules["torchvision.models.resnet"]
    return ns
