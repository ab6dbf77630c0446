"""Drop-in overlay: make `import yolov6...` resolve to the reference's package with ONLY the hot-path modules
replaced by this package (the boundary is the Python module API).

    import yolov6_amd
    yolov6_amd.install_as_yolov6("/path/to/YOLOv6")      # or with the checkout already on sys.path / as cwd
    import tools.eval                                     # the reference's own CLI, unchanged

What is substituted (everything else - core/, data/, solver/, utils/{events,config,ema,metrics,...}, the Lite /
distillation models - keeps loading from the reference checkout):

    yolov6.layers.common            -> yolov6_amd.layers.common      (+ the reference's other block classes, untouched,
                                                                      so `from yolov6.layers.common import *` stays complete)
    yolov6.models.{yolo,efficientrep,reppan,effidehead}              -> yolov6_amd.models.*
    yolov6.models.losses.{loss,loss_fuseab,loss_distill,loss_distill_ns}, yolov6.models.heads.{effidehead_fuseab,effidehead_distill_ns} -> yolov6_amd.models.*
    yolov6.assigners[.tal_assigner/.atss_assigner/.anchor_generator] -> yolov6_amd.assigners.*
    yolov6.utils.nms                -> yolov6_amd.utils.nms          (the reference's imports cv2 + torchvision at the top)
    yolov6.utils.checkpoint         -> yolov6_amd.utils.checkpoint   (same functions; torch>=2.6-safe un-pickling)
    yolov6.utils.torch_utils        -> the REFERENCE module with fuse_model / fuse_conv_and_bn / initialize_weights
                                       re-pointed at this package (time_sync, get_model_info, ... stay the reference's)

Without a reference checkout the old behaviour remains: `yolov6` aliases this package alone (enough for un-pickling
checkpoints and for code that only imports the hot-path names).
"""
import importlib
import importlib.util
import os
import sys
import types

_PKG = __name__.rsplit(".", 1)[0]          # "yolov6_amd"

# reference submodule -> replacement in this package
REPLACED = {
    "layers.common": "layers.common",
    "models.yolo": "models.yolo",
    "models.efficientrep": "models.efficientrep",
    "models.reppan": "models.reppan",
    "models.effidehead": "models.effidehead",
    "models.losses.loss": "models.losses.loss",
    "models.losses.loss_fuseab": "models.losses.loss_fuseab",
    "models.losses.loss_distill": "models.losses.loss_distill",          # core/engine.py:26, used :309-313 (--distill)
    "models.losses.loss_distill_ns": "models.losses.loss_distill_ns",    # core/engine.py:27 (n / s models: distill_ns head)
    "models.heads.effidehead_fuseab": "models.heads.effidehead_fuseab",
    "models.heads.effidehead_distill_ns": "models.heads.effidehead_distill_ns",
    "assigners": "assigners",
    "assigners.tal_assigner": "assigners.tal_assigner",
    "assigners.atss_assigner": "assigners.atss_assigner",
    "assigners.anchor_generator": "assigners.anchor_generator",
    "utils.nms": "utils.nms",
    "utils.checkpoint": "utils.checkpoint",
}
# reference modules that stay, with these names re-pointed: {submodule: {name: (our submodule, our name)}}
PATCHED = {
    "utils.torch_utils": {"fuse_model": ("utils.torch_utils", "fuse_model"),
                          "fuse_conv_and_bn": ("utils.torch_utils", "fuse_conv_and_bn"),
                          "initialize_weights": ("utils.torch_utils", "initialize_weights")},
}
# replaced modules whose reference original still provides names the rest of the reference star-imports
BACKFILLED = ("layers.common",)


def find_reference(root=None):
    """Directory of the reference's `yolov6` package: `<root>/yolov6`, else the first `yolov6/` on sys.path / cwd that
    is not this package."""
    here = os.path.dirname(os.path.abspath(__file__))
    cands = [root] if root else list(sys.path) + [os.getcwd()]
    for c in cands:
        if not c:
            c = os.getcwd()
        d = os.path.join(c, "yolov6")
        if os.path.isfile(os.path.join(d, "__init__.py")) and os.path.realpath(d) != os.path.realpath(here):
            return d
    return None


def _ours(sub):
    return importlib.import_module(f"{_PKG}.{sub}")


def _load_private(ref_dir, sub, alias):
    """Execute the reference's own source of a replaced module under a private name (its classes are plain aten
    nn.Modules outside the hot path: Lite blocks, RepOpt blocks, ...)."""
    path = os.path.join(ref_dir, *sub.split(".")) + ".py"
    spec = importlib.util.spec_from_file_location(alias, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    return mod


def install(root=None, strict=False):
    """Install the overlay.  Returns the reference package directory used (None: alias-only mode)."""
    ref_dir = find_reference(root)
    if ref_dir is None:
        if strict:
            raise ImportError("yolov6_amd: no reference checkout found (pass its root, or put it on sys.path)")
        base = sys.modules[_PKG]
        sys.modules.setdefault("yolov6", base)
        for sub in list(REPLACED.values()) + ["layers", "models", "models.losses", "models.heads", "utils", "utils.torch_utils",
                                              "utils.general"]:
            sys.modules.setdefault(f"yolov6.{sub}", _ours(sub))
        return None

    ref_root = os.path.dirname(ref_dir)
    if ref_root not in sys.path:
        sys.path.append(ref_root)                    # `import tools.eval`, `configs/*.py`
    for name in [m for m in sys.modules if m == "yolov6" or m.startswith("yolov6.")]:
        del sys.modules[name]                        # a half-imported reference package would shadow the overlay
    pkg = types.ModuleType("yolov6")
    pkg.__path__ = [ref_dir]
    pkg.__file__ = os.path.join(ref_dir, "__init__.py")
    pkg.__package__ = "yolov6"
    pkg.__y6_overlay__ = True
    sys.modules["yolov6"] = pkg

    for sub, mine in REPLACED.items():
        mod = _ours(mine)
        sys.modules[f"yolov6.{sub}"] = mod
        ref_sub = os.path.join(ref_dir, *sub.split("."))
        if hasattr(mod, "__path__") and os.path.isdir(ref_sub) and ref_sub not in mod.__path__:
            mod.__path__.append(ref_sub)             # un-replaced siblings (assigners/iou2d_calculator.py, ...) stay reachable
        parent, _, leaf = sub.rpartition(".")
        if parent:
            try:
                setattr(importlib.import_module(f"yolov6.{parent}"), leaf, mod)
            except ImportError:
                pass

    for sub, names in PATCHED.items():
        ref_mod = importlib.import_module(f"yolov6.{sub}")
        for name, (mine, attr) in names.items():
            setattr(ref_mod, name, getattr(_ours(mine), attr))

    for sub in BACKFILLED:
        mine = sys.modules[f"yolov6.{sub}"]
        try:
            orig = _load_private(ref_dir, sub, f"yolov6.{sub.rsplit('.', 1)[0]}._reference_{sub.rsplit('.', 1)[1]}")
        except Exception as e:                       # noqa: BLE001 - e.g. a dependency of the reference is absent
            if strict:
                raise
            mine.__dict__.setdefault("__y6_backfill_error__", repr(e))
            continue
        for name, val in vars(orig).items():
            if not name.startswith("_") and name not in mine.__dict__:
                setattr(mine, name, val)
    return ref_dir
