"""CPU, world_size 2, gloo: the replica plumbing bench.py uses for N > 1 GPUs."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %r)
from yolov6_amd.parallel import Replicas
r = Replicas(backend="gloo")
r.barrier()
t0 = time.perf_counter()
time.sleep(0.05 * (r.rank + 1))          # rank 1 is the slow one
r.barrier()
elapsed = r.max_over_ranks(0.1 * (r.rank + 1))
out = dict(rank=r.rank, world=r.world, shard=list(r.shard(11)), elapsed=elapsed,
           value=r.throughput(32, 5, elapsed), main=r.is_main)
print("RESULT " + json.dumps(out), flush=True)
r.close()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_replicas(tmp_path):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    res = {}
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
        r = json.loads(line[7:])
        res[r["rank"]] = r
    assert res[0]["world"] == 2 and res[0]["main"] and not res[1]["main"]
    # shards are disjoint, balanced and cover the set
    assert res[0]["shard"] == [0, 1, 2, 3, 4, 5] and res[1]["shard"] == [6, 7, 8, 9, 10]
    # both ranks agree on the slowest rank's time; whole-job throughput uses it
    assert res[0]["elapsed"] == pytest.approx(0.2) and res[1]["elapsed"] == pytest.approx(0.2)
    assert res[0]["value"] == pytest.approx(2 * 32 * 5 / 0.2)


def test_single_process_is_a_no_op():
    from yolov6_amd.parallel import Replicas
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    try:
        r = Replicas()
        assert r.world == 1 and r.is_main and list(r.shard(5)) == [0, 1, 2, 3, 4]
        assert r.max_over_ranks(1.5) == 1.5 and r.throughput(32, 10, 2.0) == 160.0
        r.barrier()
        r.close()
    finally:
        os.environ.update(env)


DDP_WORKER = r'''
import json, os, sys
sys.path.insert(0, %r)
import torch, torch.nn as nn
from yolov6_amd.parallel import Replicas, GradReducer
from yolov6_amd.train_engine import ParamArena
torch.manual_seed(0)
torch.set_num_threads(1)
net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 4, 1))
x = torch.randn(8, 3, 12, 12)
y = torch.randn(8, 4, 12, 12)
def loss_of(xs, ys):
    return ((net(xs) - ys) ** 2).mean()
# single process, concatenated batch
ref = torch.autograd.grad(loss_of(x, y), list(net.parameters()))
r = Replicas(backend="gloo")
arena = ParamArena(net, "cpu")                       # p.data / p.grad become views of two flat arrays
params = list(net.parameters())
# op i of a fake backward plan finalises parameter i of the arena (arena order = reverse registration order)
marks = [(i + 1, [p]) for i, p in enumerate(arena.params)]
calls = []
class FakeGraph:
    arena, bwd_marks, n_bwd_ops = arena, marks, len(arena.params)
    def backward(self, grads, first=0, last=None):
        calls.append((first, last))
        if first == 0:                               # the shard's gradient (autograd accumulates into the arena views)
            sh = r.shard(8)
            arena.zero_grad()
            loss_of(x[sh.start:sh.stop], y[sh.start:sh.stop]).backward()
red = GradReducer(arena, marks, len(arena.params), r, chunks=3, average=True)
red.run_backward(FakeGraph(), None)
err = max(float((p.grad - g).abs().max()) for p, g in zip(params, ref))
cover = sorted((lo, hi) for _, _, lo, hi in red.segments)
print("RESULT " + json.dumps(dict(rank=r.rank, err=err, calls=calls, segments=red.segments, numel=arena.numel,
                                  views=all(p.grad.data_ptr() == arena.grad.data_ptr() + 4 * arena.offset_of(p) for p in params))), flush=True)
r.close()
'''

def test_forced_one_rank_group(monkeypatch):
    """Y6_FORCE_DIST=1: a one-rank job brings the process group up (gloo here, RCCL on a GPU box) and runs the barrier / MAX
    reduce / gradient all-reduce code of an N-rank launch - the path `tools/gpu_visit_r03t.sh` exercises on one MI355X."""
    import torch
    from yolov6_amd.parallel import GradReducer, Replicas
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("Y6_FORCE_DIST", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29571")
    r = Replicas(backend="gloo")
    try:
        assert r.world == 1 and r.dist is not None and r.dist.is_initialized()
        r.barrier()
        assert r.max_over_ranks(2.5) == 2.5

        class _Arena:
            pass
        a = _Arena()
        p = torch.nn.Parameter(torch.zeros(8))
        a.params, a.offsets, a.numel, a.grad = [p], [0], 8, torch.arange(8, dtype=torch.float32)
        red = GradReducer(a, [(1, [p])], 1, r, chunks=1, average=True)
        red.reduce_range(0, 8)
        assert torch.equal(a.grad, torch.arange(8, dtype=torch.float32))     # sum over one rank, / 1
    finally:
        r.close()
    assert r.dist is None



def test_two_rank_gloo_gradient_exchange_equals_concatenated_batch():
    """DDP semantics of the training step (engine.py:455-468) on the arena: per-rank shard gradients, chunked all-reduce
    interleaved with the (here: fake) backward plan, average == the single-process gradient of the concatenated batch."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", DDP_WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
        assert res["err"] < 1e-6, res
        assert res["views"]
        segs = res["segments"]
        # the segments tile the backward plan and the arena without gaps
        assert segs[0][0] == 0 and segs[-1][1] == 6 and all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
        assert segs[0][2] == 0 and segs[-1][3] == res["numel"] and all(a[3] == b[2] for a, b in zip(segs, segs[1:]))
        assert [c[0] for c in res["calls"]] == [s[0] for s in segs if s[1] > s[0] or s[0] == 0]


SEAM_WORKER = r"""
import json, os, sys
sys.path.insert(0, %r)
MODE = %r
import torch, torch.nn as nn, torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP
from yolov6_amd.parallel import Replicas, install_grad_reducer
from yolov6_amd.train_engine import ParamArena, run_train_graph
torch.manual_seed(0)
torch.set_num_threads(1)


class StandInGraph:
    # What train_engine.TrainGraph is to the model, with torch CPU ops in place of the native plans: forward without
    # autograd, backward writes the parameter gradients straight into the arena views (never through autograd).
    def __init__(self, model):
        self.model, self.arena = model, ParamArena(model, "cpu")
        self.anchor = torch.zeros((), requires_grad=True)
        self.bwd_marks = [(i + 1, [p]) for i, p in enumerate(self.arena.params)]
        self.n_bwd_ops = len(self.arena.params)
    def forward(self, x):
        self.x = x
        with torch.no_grad():
            return (self.model.net(x),)
    def backward(self, grads, first=0, last=None):
        if first != 0:
            return
        if self.arena.params[0].grad is None:
            self.arena.zero_grad(); self.arena.reattach()
        with torch.enable_grad():
            gs = torch.autograd.grad(self.model.net(self.x), list(self.model.net.parameters()), grads[0])
        for p, g in zip(self.model.net.parameters(), gs):
            p.grad.add_(g)                               # accumulate into the arena view, as the kernels do


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 4, 1))
    def forward(self, x):
        g = self.__dict__.get("_graph")
        if g is None:
            g = self.__dict__["_graph"] = StandInGraph(self)
        return run_train_graph(self, g, x)[0]


x, y = torch.randn(8, 3, 12, 12), torch.randn(8, 4, 12, 12)
ref_net = Net()
ref = torch.autograd.grad(((ref_net.net(x) - y) ** 2).mean(), list(ref_net.net.parameters()))
r = Replicas(backend="gloo")
model = Net()
model.load_state_dict(ref_net.state_dict())
if MODE == "ddp":
    wrapped = DDP(model)                                 # what core/engine.py:466 does to the model
else:
    wrapped = model
    install_grad_reducer(model, r, chunks=3)
sh = r.shard(8)
errs = []
for it in range(3):                                      # several iterations: DDP re-arms its reducer every forward
    for p in model.parameters():
        p.grad = None                                    # optimizer.zero_grad(set_to_none=True)
    out = wrapped(x[sh.start:sh.stop])
    ((out - y[sh.start:sh.stop]) ** 2).mean().backward()
    errs.append(max(float((p.grad - g).abs().max()) for p, g in zip(model.net.parameters(), ref)))
arena = model.__dict__["_graph"].arena
views = all(p.grad.data_ptr() == arena.grad.data_ptr() + 4 * arena.offset_of(p) for p in model.net.parameters())
local = torch.autograd.grad(((ref_net.net(x[sh.start:sh.stop]) - y[sh.start:sh.stop]) ** 2).mean(), list(ref_net.net.parameters()))
differs = max(float((l - g).abs().max()) for l, g in zip(local, ref))
print("RESULT " + json.dumps(dict(rank=r.rank, errs=errs, views=views, local_vs_global=differs)), flush=True)
r.close()
"""


@pytest.mark.parametrize("mode", ["ddp", "reducer"])
def test_two_rank_gloo_training_seam(mode):
    """The training drop-in seam on two ranks: (ddp) the module WRAPPED IN torch DistributedDataParallel, exactly as the
    reference's trainer does (core/engine.py:455-468), although its backward writes `p.grad` outside autograd - the zero
    gradients _TrainStepFn threads through autograd make DDP's hooks fire; (reducer) the bare module with the native
    GradReducer installed.  Either way every rank ends with the gradient of the concatenated batch, in the arena views."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", SEAM_WORKER % (ROOT, mode)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
        assert res["local_vs_global"] > 1e-3, res          # the shards really differ: an un-reduced gradient would fail below
        assert max(res["errs"]) < 1e-6, res
        assert res["views"], res


def test_bench_train_two_ranks_self_spawn_on_gloo(tmp_path):
    """`python bench.py --mode train --gpus 2` as the driver will launch it on an 8-GPU node, minus the GPUs (Y6_BENCH_MOCK=1:
    CPU ranks over gloo, a stand-in step through parallel.GradReducer): the file re-executes itself under torch.distributed.run
    with a free port on 127.0.0.1, both ranks meet, rank 0 prints ONE JSON line and it is the LAST line of stdout (a banner
    precedes it), n_gpus = 2, the global batch doubles, the gradient is the all-reduced one.  Two launches at once must not
    collide on the rendezvous port."""
    env = dict(os.environ, Y6_BENCH_MOCK="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", "8"]
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path)) for _ in range(2)]
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-3000:]
        outs.append(out)
    one = subprocess.run([c if c != "2" else "1" for c in cmd], env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert one.returncode == 0, one.stderr[-3000:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    for out in outs:
        lines = [l for l in out.strip().splitlines() if l.strip()]
        assert sum(1 for l in lines if l.startswith("{")) == 1, lines          # one JSON line (rank 0 only)
        r = json.loads(lines[-1])                                              # ... and it is the last line
        assert r["n_gpus"] == 2 and r["steps"] == 5 and r["warmup"] == 2
        assert r["config"]["global_batch"] == 2 * r1["config"]["global_batch"] == 16
        assert r["config"]["parallelism"] == "dp2" and r["scaling"] == "weak"
        assert r["grad_abs_sum"] != pytest.approx(r1["grad_abs_sum"], rel=1e-9)   # the average over two different shards


PENDING_WORKER = r"""
import json, os, sys
sys.path.insert(0, %r)
import torch
from yolov6_amd.parallel import GradReducer, Replicas
from yolov6_amd.train_engine import ParamArena
r = Replicas(backend="gloo")
net = torch.nn.Linear(4, 4)
arena = ParamArena(net, "cpu")
marks = [(i + 1, [p]) for i, p in enumerate(arena.params)]
class Plan:
    pending = 0
    def side_pending(self):
        return self.pending
class Graph:
    bwd_marks, n_bwd_ops, bwd_plan = marks, len(arena.params), Plan()
    def backward(self, grads, first=0, last=None):
        if first == 0:
            arena.zero_grad()
            net(torch.ones(2, 4)).sum().backward()
Graph.arena = arena
red = GradReducer(arena, marks, len(arena.params), r, chunks=2, average=True)
g = Graph()
red.run_backward(g, None)                       # joined side stream: fine
ok_first = True
Graph.bwd_plan.pending = 3                      # a plan that returned with side-stream work the launch stream is not ordered behind
try:
    red.run_backward(g, None)
    refused = False
except RuntimeError as e:
    refused = "side stream" in str(e)
print("RESULT " + json.dumps(dict(rank=r.rank, ok_first=ok_first, refused=refused)), flush=True)
r.close()
"""


def test_two_rank_gloo_reducer_refuses_an_unjoined_side_stream():
    """parallel.GradReducer hands a gradient chunk to the all-reduce behind an event recorded on the launch stream; the backward
    plan writes weight gradients on its side stream.  The plan joins before it returns (csrc/plan.hip run_ops,
    y6_plan_side_pending == 0); the reducer asserts it - a plan that did not would let the all-reduce read half-written
    gradients.  Both ranks refuse (nobody is left waiting in a collective)."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", PENDING_WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
        assert res["ok_first"] and res["refused"], res


BUFFER_WORKER = r"""
import json, os, sys
sys.path.insert(0, %r)
import torch, torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP
from yolov6_amd.parallel import Replicas
from yolov6_amd.train_engine import ParamArena
torch.manual_seed(0)
torch.set_num_threads(1)
r = Replicas(backend="gloo")
net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU())
arena = ParamArena(net, "cpu")                     # parameters become views of the flat arena; buffers stay what they are
bn = net[1]
ptrs0 = (bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr())
seen = []
net.register_forward_pre_hook(lambda m, a: seen.append(bn.running_mean.clone()))
ddp = DDP(net)                                     # broadcast_buffers=True: rank 0's buffers are written INTO every rank's, in place
g = torch.Generator().manual_seed(10 + r.rank)
for it in range(3):
    x = torch.randn(4, 3, 8, 8, generator=g) + r.rank     # different statistics per rank
    ddp(x).sum().backward()
ptrs1 = (bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr())
# what rank 0 had at the start of forward k is what every rank sees at the start of forward k
mine = torch.stack(seen)
ref = mine.clone()
r.dist.broadcast(ref, src=0)
params_are_views = all(p.data_ptr() == arena.data.data_ptr() + 4 * arena.offset_of(p) for p in net.parameters())
print("RESULT " + json.dumps(dict(rank=r.rank, same_storage=ptrs0 == ptrs1, follows_rank0=bool(torch.equal(mine, ref)),
                                  moved=float((mine[-1] - mine[0]).abs().max()), params_are_views=params_are_views)), flush=True)
r.close()
"""


def test_two_rank_gloo_ddp_buffer_broadcast_lands_in_the_storage_the_plans_hold():
    """BatchNorm running statistics under a torch-DDP wrapper (reference core/engine.py:463-466 wraps the model; DDP broadcasts
    rank 0's buffers into every rank's at each forward).  The native statistics op holds the ADDRESS of running_mean / running_var
    (the training graph is built once): the broadcast must be an in-place write into that storage, and after it every rank
    must see rank 0's statistics.  (The native GradReducer path does not broadcast: statistics stay per rank and rank 0's are the
    ones a checkpoint keeps - the same end state as DDP's.)"""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", BUFFER_WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
        res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
        assert res["same_storage"] and res["follows_rank0"] and res["params_are_views"], res
        assert res["moved"] > 1e-4, res


def test_bench_infer_two_ranks_self_spawn_on_gloo(tmp_path):
    """`python bench.py --gpus 2` (the headline, inference mode) as the driver launches it on a multi-GPU node, minus the GPUs
    (Y6_BENCH_MOCK=1): self-spawn under torch.distributed.run on 127.0.0.1, ranks whose set-up takes DIFFERENT time meet in the
    timed windows' barriers (no hang), `timed_window` - the function the GPU path times with - runs EXACTLY K steps per window on every
    rank, rank 0 prints ONE JSON line last: n_gpus 2, replicas (no collective), weak scaling, global batch doubled."""
    env = dict(os.environ, Y6_BENCH_MOCK="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert sum(1 for l in lines if l.startswith("{")) == 1, lines
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 64 and "no collective" in d["config"]["parallelism"]
    assert d["steps_enqueued_rank0"] == 7 * 4            # three windows + the one-at-a-time window, exactly K steps each
    assert d["value"] > 0 and d["sequential"]["value"] > 0
    assert d["supervisor"]["attempts"] == 1
