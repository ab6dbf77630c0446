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
