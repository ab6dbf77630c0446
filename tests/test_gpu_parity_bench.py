"""GPU: parity AT THE BENCHMARKED CONFIGURATIONS (BASELINE configs[1] and [3]).

* YOLOv6-S deploy, 640x640, batch 32, the autotuned plan exactly as `bench.py` builds and times it (same synthetic
  weights, same calibrated head bias, same kernel variants per layer), and
* YOLOv6-L6 at FULL width, 1280x1280 (batch 2: the per-image work is identical at batch 8; the CPU oracle stays in
  seconds),

each checked (a) per layer, teacher-forced: every conv / convT / SPPF / decode op of the plan runs alone on the
fp16-emulating ORACLE's activation of the previous layer and must match the oracle's output within the north_star's
1e-3 (`max |hip - ref| / max(1,|ref|)`); (b) end to end: class scores within 1e-3 absolute, boxes reported in
ABSOLUTE PIXELS together with the layer after which the free-running pipelines differ most.
The per-layer table is written to gpurun_out/parity_<model>.json for DESIGN.md.
"""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from oracle.model_oracle import Oracle
from tests.plan_replay import OracleChain, box_report

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# measured on MI355X (profiles/r02/parity_*.json) + 20 %: worst absolute box deviation, in input pixels, between the HIP
# pipeline and the fp16-emulating oracle running freely (no teacher forcing) on the same input
BOX_PX_BOUND = {"yolov6s": 0.1}          # measured 0.078 px (one fp16 ulp of a regression distance x stride 32 ... 64)


def _bench_setup(name, size, batch):
    import bench
    args = argparse.Namespace(model=name, size=size, batch=batch)
    cfg, sd, model, x = bench.build_model_and_input(args, torch.device(DEV))
    shift = bench.calibrate_head_bias(model, x)
    sd = {k: (v + shift if ("cls_preds" in k and k.endswith(".bias")) else v) for k, v in sd.items()}
    return cfg, sd, model, x


@pytest.mark.parametrize("name,size,batch", [("yolov6s", 640, 32), ("yolov6l6", 1280, 2)])
def test_per_layer_teacher_forced_and_end_to_end(name, size, batch):
    cfg, sd, model, x = _bench_setup(name, size, batch)
    plan = model.compile(x, autotune=True)           # what bench.py times
    det_hip = plan.run().clone()
    torch.cuda.synchronize()
    orc = Oracle(cfg, sd, 80, emulate_fp16=True)
    with torch.no_grad():
        ref, _ = orc.forward(x.float().cpu())
        chain = OracleChain(plan, orc)
        # free-running comparison first (HIP buffers still hold the HIP pipeline's own activations)
        chain.run(teacher_force=False)
        # the chain multiplies with the plan's folded weights, Oracle.forward with its own fold of the same parameters
        # (same formulas, fp32 association differs in the bias sum): equal up to isolated fp16 flips
        sync = float(((chain.final - ref).abs() / ref.abs().clamp(min=1.0)).max())
        assert sync < 2e-3, f"per-layer oracle chain deviates from Oracle.forward by {sync:.3e} (test harness out of sync)"
        free = []
        for i, e in enumerate(plan.op_log):
            outs = e.get("outs") or ([e["out"]] if e["kind"] in ("conv", "stem", "convt", "pw_s2", "stem_s2") else [])
            for r in outs:
                v = chain._get(r)
                d = (chain._download(r) - v).abs()
                free.append(dict(op=i, err=float((d / v.abs().clamp(min=1.0)).max())))
        chain2 = OracleChain(plan, orc)
        rows = chain2.run(teacher_force=True)
    variants = {r["op"]: r["variant"] for r in plan.timing_read() if r["variant"]}
    with torch.no_grad():
        ref32, _ = Oracle(cfg, sd, 80, emulate_fp16=False).forward(x.float().cpu())
    floor = box_report(ref.numpy(), ref32.numpy())          # fp16-emulating oracle vs fp32 oracle: the reference's own fp16 noise
    worst = max(rows, key=lambda r: r["err"])
    rep = box_report(det_hip.cpu().numpy(), ref.numpy())
    jump, prev = dict(op=-1, gain=0.0), 0.0
    for f in free:
        if f["err"] - prev > jump["gain"]:
            jump = dict(op=f["op"], gain=f["err"] - prev, err=f["err"])
        prev = max(prev, f["err"])
    desc = {r["op"]: r["desc"] for r in rows}
    summary = dict(model=name, size=size, batch=batch, ops=len(rows), chain_vs_oracle_forward=sync, per_layer_max=worst["err"], per_layer_worst=worst["desc"],
                   end_to_end=rep, reference_fp16_vs_fp32=floor, free_running_largest_jump=dict(jump, desc=desc.get(jump["op"], "?")),
                   free_running_final_layer_err=free[-1]["err"] if free else None)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"parity_{name}_{size}_b{batch}.json"), "w") as f:
        json.dump(dict(summary=summary, rows=rows, free_running=free, variants=variants), f, indent=1)
    print(json.dumps(summary))
    # per-op bound: 1e-3 (conv + bias + ReLU / none, decode); 2 fp16 ulp for SiLU, 3 with the residual add (see
    # tests/gpu_utils.py::op_tolerance - both sides round at the same op boundaries, fp32 accumulation order flips isolated
    # roundings by one ulp and SiLU / the shortcut add re-round them)
    from tests.gpu_utils import op_tolerance
    bad = []
    for r in rows:
        d = r["desc"]
        act = "silu" if " silu" in d else ("hardswish" if "hardswish" in d else "relu")
        tol = op_tolerance(act, with_res="+res" in d)
        r["tol"] = tol
        if r["kind"] == "pred_decode":     # conv + decode in one op: scores at the conv's bar, boxes at one fp16 ulp of the distance x stride
            if not (r["err_scores"] <= 1e-3 and r["err_box_px"] <= r["box_tol_px"]):      # (NaN is a failure)
                bad.append(r)
        elif not (r["err"] <= tol):
            bad.append(r)
    assert not bad, f"{name}: {len(bad)} ops above their bound teacher-forced, e.g. {bad[:3]}"
    if name == "yolov6s":
        assert rep["scores_max"] <= 1e-3, f"{name}: class scores deviate by {rep['scores_max']:.3e} end to end"
        assert rep["max_px"] <= BOX_PX_BOUND[name], f"{name}: boxes deviate by {rep['max_px']:.3f} px end to end"
    else:
        # 204 fp16-stored ops deep on random weights: free-running pipelines drift apart by amplified rounding flips; the
        # scale is the reference's OWN fp16 deviation from its fp32 result on the same input
        assert rep["scores_max"] <= 2.0 * floor["scores_max"] + 1e-3, (rep, floor)
        assert rep["p999_px"] <= 2.0 * floor["p999_px"] + 0.1, (rep, floor)


def test_bench_step_nms_equals_oracle_nms():
    """What bench.py times (plan.run + nms_raw on the b32 tensor): NMS output == oracle NMS of the same `det`."""
    from oracle import nms_oracle
    from yolov6_amd.utils.nms import nms_raw
    import bench
    cfg, sd, model, x = _bench_setup("yolov6s", 640, 32)
    plan = model.compile(x, autotune=True)
    det = plan.run()
    dets, index, count = nms_raw(det, bench.CONF, bench.IOU, multi_label=True, max_det=bench.MAX_DET)
    torch.cuda.synchronize()
    bench.verify_nms(det, dets, index, count, images=range(0, 32, 4))


def test_decode_launch_candidates_equal_nms_first_stage():
    """Plan.attach_nms (include/yolov6_hip.h y6_nms_sink): the fused head tail selects the NMS candidates of its rows while they
    are in LDS.  The detections, flat indices and counts must equal - bit for bit - those of y6_nms selecting them itself from
    the prediction tensor, for multi-label / best-class / a class filter, on repeated runs (the per-image key lists are re-zeroed
    by every decode launch), and the full-size case must equal the oracle."""
    from yolov6_amd.utils.nms import nms_raw
    import bench
    cfg, sd, model, x = _bench_setup("yolov6s", 640, 32)
    plan = model.compile(x, autotune=False)
    for kw in (dict(conf_thres=bench.CONF, multi_label=True, classes=None), dict(conf_thres=0.25, multi_label=False, classes=None),
               dict(conf_thres=0.1, multi_label=True, classes=[0, 3, 17, 79])):
        plan.attach_nms(None)
        det = plan.run()
        ref = nms_raw(det, kw["conf_thres"], bench.IOU, classes=kw["classes"], multi_label=kw["multi_label"], max_det=bench.MAX_DET)
        torch.cuda.synchronize()
        ref = [t.clone() for t in ref]
        tok = plan.attach_nms(kw["conf_thres"], kw["classes"], kw["multi_label"])
        assert tok is not None, "the YOLOv6-S plan has a fused head tail: the sink must attach"
        for rep in range(3):
            det2 = plan.run()
            got = nms_raw(det2, kw["conf_thres"], bench.IOU, classes=kw["classes"], multi_label=kw["multi_label"], max_det=bench.MAX_DET,
                          candidates=tok)
            torch.cuda.synchronize()
            assert torch.equal(det2, det)
            for a, b, nm in zip(got, ref, ("dets", "index", "count")):
                assert torch.equal(a, b), f"{nm} differ with candidates from the decode launch ({kw}, run {rep})"
        assert int(ref[2].sum()) > 0 or kw["classes"] is not None or not kw["multi_label"]
        with pytest.raises(RuntimeError):   # other thresholds than the sink's: refused, not silently wrong
            nms_raw(det2, kw["conf_thres"] + 0.01, bench.IOU, classes=kw["classes"], multi_label=kw["multi_label"], candidates=tok)
    tok = plan.attach_nms(bench.CONF, None, True)
    det = plan.run()
    dets, index, count = nms_raw(det, bench.CONF, bench.IOU, multi_label=True, max_det=bench.MAX_DET, candidates=tok)
    torch.cuda.synchronize()
    bench.verify_nms(det, dets, index, count, images=(0, 13, 31))


def test_inflight_runner_equals_one_at_a_time():
    """yolov6_amd.pipeline.InflightRunner (two batches in flight on two HIP streams, a plan each): the detections of every
    batch equal those of the plain `model(x)` + `non_max_suppression` path, batch by batch, for a sequence of different batches -
    nothing of one batch leaks into the other (own activation buffers, own NMS workspace, own result tensors)."""
    from oracle import synth
    from yolov6_amd.pipeline import InflightRunner
    import bench
    cfg, sd, model, x = _bench_setup("yolov6s", 320, 8)
    batches = [synth.synth_images(8, 320, seed=40 + i).to(x.device).half() for i in range(5)]
    # one at a time: the same runner with one plan, every result consumed before the next batch is submitted (both runners take the
    # shape-derived kernel variants, so that the two paths run the same kernels - a timed choice is per plan)
    one = InflightRunner(model, batches[0].clone(), depth=1, conf_thres=bench.CONF, iou_thres=bench.IOU, multi_label=True,
                         max_det=bench.MAX_DET, autotune=False)
    want = [[t.clone() for t in one.submit(b).result()[0]] for b in batches]
    run = InflightRunner(model, batches[0].clone(), depth=2, conf_thres=bench.CONF, iou_thres=bench.IOU, multi_label=True,
                         max_det=bench.MAX_DET, autotune=False)
    tickets = [run.submit(b) for b in batches[:2]]
    got = []
    for b in batches[2:]:
        got.append(tickets.pop(0).result()[0])
        got[-1] = [t.clone() for t in got[-1]]
        tickets.append(run.submit(b))
    for t in tickets:
        got.append([u.clone() for u in t.result()[0]])
    stale = run.submit(batches[0])
    run.submit(batches[1])
    run.submit(batches[2])                               # reuses the slot of `stale`
    with pytest.raises(RuntimeError, match="slot was reused"):
        stale.result()
    assert len(got) == len(want) == 5
    for k, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w) == 8
        for a, b_ in zip(g, w):
            assert torch.equal(a, b_), f"batch {k}: detections differ from the one-at-a-time path"
    assert sum(int(t.shape[0]) for w in want for t in w) > 0


def test_whole_step_tuned_plan_and_its_copy_run_the_same_kernels():
    """y6_plan_autotune's whole-step pass leaves every conv op with a named variant the op supports; `new_plan(variants_from=...)`
    (the in-flight slots: tuned once) takes the table over without tuning again - same variant table, same hash - and, running the
    same kernels on the same input, produces the same BITS; the tuned plan agrees with the shape-derived plan of the same model
    within the fp16 tolerance of a different accumulation order."""
    cfg, sd, model, x = _bench_setup("yolov6s", 320, 8)
    tuned = model.compile(x, autotune=True)
    tab = tuned.variant_table()
    assert tab and all(name not in ("", "shape-derived", "naive") for _, name in tab), tab
    twin = model.new_plan(x, autotune=True, variants_from=tuned)
    assert twin.variant_table() == tab and twin.variant_hash() == tuned.variant_hash()
    a = tuned.run().clone()
    b = twin.run().clone()
    torch.cuda.synchronize()
    assert torch.equal(a, b), "two plans with the same kernel table must produce the same bits"
    shape = model.new_plan(x, autotune=False)
    c = shape.run().clone()
    torch.cuda.synchronize()
    assert float((a[..., 5:] - c[..., 5:]).abs().max()) < 5e-3          # (a sanity bound, not a parity claim: other kernels, other
    assert float((a[..., :4] - c[..., :4]).abs().max()) < 1.0           #  fp32 summation orders; parity is against the oracle elsewhere)
    other = model.new_plan(x[:4].contiguous(), autotune=False)
    with pytest.raises(RuntimeError, match="differs in shape|differ in length"):
        other.copy_variants_from(tuned)


@pytest.mark.parametrize("mode", ["infer", "train"])
def test_bench_under_torchrun_with_a_one_rank_rccl_group(mode):
    """The driver's N > 1 launch form - `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` - with the one GPU
    a test box has and the process group forced up (Y6_FORCE_DIST=1): the RCCL communicator, the barriers of timed_window, the MAX
    all-reduce and (train) the chunked gradient all-reduce behind the backward plan all execute; rank 0 prints one JSON line."""
    import subprocess
    import sys
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    extra = ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-train-sub", "--no-config-subs", "--dropin-steps", "0"] if mode == "infer" else \
        ["--mode", "train", "--steps", "3", "--warmup", "2"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    env = dict(os.environ, Y6_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    if mode == "infer":
        assert d["inflight"] == 2 and d["sequential"]["value"] > 0 and d["self_check"]["nms_equals_oracle_images"] >= 2
