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
