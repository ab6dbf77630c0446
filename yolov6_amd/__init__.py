"""yolov6_amd - the MI355X (gfx950) native hot path of YOLOv6 behind the reference's
Python module API (yolov6.layers / yolov6.models / yolov6.assigners / yolov6.utils.nms).

    from yolov6_amd import build_model, get_config, non_max_suppression
    model = build_model(get_config("yolov6s"), 80, "cuda").eval().half()
    det, _ = model(images)                       # HIP kernels, one native plan
    boxes = non_max_suppression(det, 0.03, 0.65, multi_label=True)

`install_as_yolov6()` registers the package under the reference's import paths so that the
reference's callers (tools/eval.py, core/evaler.py, checkpoints that pickle
`yolov6.models.yolo.Model`) resolve to these classes - see INTEGRATION.md.
"""
import sys

from .configs import Config, get_config, load_config  # noqa: F401


def __getattr__(name):
    # lazy: importing the package must not need torch-heavy modules or the .so
    if name in ("build_model", "Model"):
        from .models import yolo
        return getattr(yolo, name)
    if name == "non_max_suppression":
        from .utils.nms import non_max_suppression
        return non_max_suppression
    if name == "TaskAlignedAssigner":
        from .assigners import TaskAlignedAssigner
        return TaskAlignedAssigner
    raise AttributeError(name)


def install_as_yolov6(reference_root=None, strict=False):
    """Make `import yolov6...` resolve to the reference checkout with the hot-path modules replaced by this package
    (overlay; see yolov6_amd/dropin.py).  Without a checkout, `yolov6` aliases this package alone.  Returns the
    reference package directory in use, or None."""
    from . import dropin
    return dropin.install(reference_root, strict=strict)
