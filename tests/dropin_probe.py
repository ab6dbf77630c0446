#!/usr/bin/env python3
"""Subprocess body of tests/test_dropin_cpu.py (build container only: needs the reference checkout).

    dropin_probe.py write <ref_root> <ckpt>   pure reference: build a small Model from the reference's own
                                              configs/yolov6s.py, synthetic weights, save a reference-format checkpoint
                                              ({'model': ..., 'ema': ...} of pickled module objects, engine.py:192-200)
    dropin_probe.py load  <ref_root> <ckpt>   overlay: install_as_yolov6(ref_root), import the reference's tools/eval.py,
                                              load the checkpoint through the reference's Evaler.init_model, print a JSON
                                              report
The stubs stand in for packages that are absent from this image and are NOT part of the hot path (cv2, torchvision,
addict, pycocotools, thop) - the same stand-ins tests/golden/gen_golden.py uses.
"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
        return sys.modules[name]

    mod("cv2", setNumThreads=lambda n: None, INTER_LINEAR=1, INTER_AREA=3, BORDER_CONSTANT=0)
    ops = mod("torchvision.ops", nms=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("torchvision stub")))
    mod("torchvision", ops=ops)

    class Dict(dict):                       # the small part of addict.Dict that yolov6/utils/config.py relies on
        def __init__(self, *a, **k):
            super().__init__()
            for key, v in dict(*a, **k).items():
                self[key] = v

        def __setitem__(self, key, v):
            super().__setitem__(key, type(self)(v) if isinstance(v, dict) and not isinstance(v, Dict) else v)

        def __getattr__(self, key):
            try:
                return self[key]
            except KeyError:
                return self.__missing__(key)

        def __missing__(self, key):
            raise KeyError(key)

        __setattr__ = __setitem__

    mod("addict", Dict=Dict)
    coco = mod("pycocotools.coco", COCO=object)
    cocoeval = mod("pycocotools.cocoeval", COCOeval=object)
    mod("pycocotools", coco=coco, cocoeval=cocoeval)
    mod("thop", profile=lambda model, inputs=(), verbose=False: (0.0, float(sum(p.numel() for p in model.parameters()))))


def small_cfg(ref_root):
    from yolov6_amd.configs import load_config
    cfg = load_config(os.path.join(ref_root, "configs", "yolov6s.py"))
    cfg.model["width_multiple"], cfg.model["depth_multiple"] = 0.125, 0.17
    return cfg


def write(ref_root, path):
    import torch
    install_stubs()
    sys.path.insert(0, ref_root)
    from copy import deepcopy
    from yolov6.models.yolo import build_model          # the REFERENCE's
    from yolov6_amd.utils import synth                   # seeds only
    import yolov6.models.yolo as ref_yolo
    assert ref_yolo.__file__.startswith(ref_root)
    model = build_model(small_cfg(ref_root), 80, "cpu")
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=0))
    ema = None if os.environ.get("Y6_PROBE_NO_EMA") else deepcopy(model).half()      # the committed fixture keeps one copy
    ckpt = {"model": deepcopy(model).half(), "ema": ema, "updates": 7, "optimizer": None, "epoch": 3}
    torch.save(ckpt, path)
    x = synth.synth_images(2, 64, seed=1)
    model.eval()
    with torch.no_grad():
        det = model(x)[0]
    print(json.dumps(dict(keys=len(model.state_dict()), det_sum=float(det.double().sum()))))


def load(ref_root, path):
    import torch
    install_stubs()
    import yolov6_amd
    used = yolov6_amd.install_as_yolov6(ref_root, strict=True)
    os.chdir(ref_root)                                   # tools/eval.py appends os.getcwd() to sys.path
    import tools.eval as ref_eval                        # the reference's CLI module, unchanged
    from yolov6.core.evaler import Evaler
    from yolov6.utils.config import Config               # reference module (addict-backed)
    from yolov6.utils.events import LOGGER               # noqa: F401 - reference module
    from yolov6.utils.general import increment_name, check_img_size
    from yolov6.layers.common import DetectBackend, RepVGGBlock
    import yolov6.layers.common as lc
    import yolov6.models.yolo as ym
    import yolov6.utils.torch_utils as tu
    import yolov6.utils.checkpoint as ck
    from yolov6_amd.models.yolo import Model as OurModel
    from yolov6_amd.layers import common as our_common

    rep = dict(ref_dir=used, tools_eval=ref_eval.__file__, evaler=sys.modules["yolov6.core.evaler"].__file__)
    rep["check_img_size"] = check_img_size(641, 32)
    rep["increment_name"] = str(increment_name("/nonexistent/exp"))
    rep["common_is_ours"] = lc is our_common
    rep["yolo_is_ours"] = ym.Model is OurModel
    rep["fuse_model_is_ours"] = tu.fuse_model.__module__
    rep["time_sync_is_reference"] = tu.time_sync.__module__
    rep["checkpoint_is_ours"] = ck.__name__
    # the four ComputeLoss classes core/engine.py:24-27 imports (its module body also needs cv2 / tensorboard / the data
    # loaders, which are outside the hot path: import the four names the way that file does)
    from yolov6.models.losses.loss import ComputeLoss as CL
    from yolov6.models.losses.loss_fuseab import ComputeLoss as CL_ab
    from yolov6.models.losses.loss_distill import ComputeLoss as CL_distill
    from yolov6.models.losses.loss_distill_ns import ComputeLoss as CL_distill_ns
    rep["loss_modules"] = [c.__module__ for c in (CL, CL_ab, CL_distill, CL_distill_ns)]
    try:                                                 # and through the trainer module itself when its imports resolve
        sys.modules.setdefault("torch.utils.tensorboard", types.ModuleType("torch.utils.tensorboard")).__dict__.setdefault("SummaryWriter", object)
        import yolov6.core.engine as eng
        rep["engine_loss_modules"] = [getattr(eng, n).__module__ for n in
                                      ("ComputeLoss", "ComputeLoss_ab", "ComputeLoss_distill", "ComputeLoss_distill_ns")]
    except Exception as e:                               # noqa: BLE001 - reported, the test decides
        rep["engine_import_error"] = f"{type(e).__name__}: {e}"
    rep["backfilled"] = [n for n in ("RealVGGBlock", "LinearAddBlock", "Lite_EffiBlockS1", "MBLABlock") if hasattr(lc, n)]
    # the reference's Evaler.init_model, on CPU (no warm-up forward there): load_checkpoint -> fuse -> switch_to_deploy
    ev = Evaler.__new__(Evaler)
    ev.device, ev.half, ev.img_size = torch.device("cpu"), False, 64
    model = ev.init_model(None, path, "val")
    rep["model_type"] = f"{type(model).__module__}.{type(model).__name__}"
    rep["stride"] = ev.stride
    foreign = sorted({type(m).__module__ for m in model.modules()
                      if not type(m).__module__.startswith(("yolov6_amd", "torch.nn"))})
    rep["foreign_module_classes"] = foreign
    rep["deployed"] = all(hasattr(m, "rbr_reparam") for m in model.modules() if isinstance(m, RepVGGBlock))
    rep["n_repvgg"] = sum(isinstance(m, RepVGGBlock) for m in model.modules())
    rep["fused"] = not any(hasattr(m, "bn") for m in model.modules() if type(m) is our_common.ConvModule)
    rep["act_names"] = sorted({str(m._activation_name()) for m in model.modules() if type(m) is our_common.ConvModule})
    be = DetectBackend(path, device=torch.device("cpu"))
    rep["backend_model_type"] = type(be.model).__name__
    rep["backend_stride"] = be.stride
    # the deploy-form state_dict that will be lowered equals the oracle's deploy transform of the same weights
    from oracle.model_oracle import deploy_state_dict
    from yolov6_amd.utils import synth
    sd0 = synth.synth_state_dict({k: v.float() for k, v in
                                  torch.load(path, weights_only=False)["model"].state_dict().items()}, seed=0)
    want = deploy_state_dict(small_cfg(ref_root), {k: v.half().float() for k, v in sd0.items()}, 80)
    got = model.state_dict()
    rep["deploy_keys_equal"] = sorted(k for k in got if "num_batches" not in k) == sorted(k for k in want if "num_batches" not in k)
    rep["deploy_max_diff"] = max(float((got[k].float() - want[k].float()).abs().max()) for k in want if k in got and "num_batches" not in k)
    cfg = Config.fromfile(os.path.join(ref_root, "configs", "yolov6s.py"))
    rep["config_type"] = cfg.model.type
    print(json.dumps(rep))


if __name__ == "__main__":
    {"write": write, "load": load}[sys.argv[1]](sys.argv[2], sys.argv[3])
