"""Multi-GPU plumbing for the hot path: one process per GPU, independent replicas.

Inference shards by independent images: every rank owns its own batch and its own
native plan; there is NO data-path collective.  The process group (backend "nccl" = RCCL on ROCm,
"gloo" on CPU) is used only for barriers and one MAX all-reduce of the elapsed time, as the bench
contract requires.

Training has the one exchange step of the path: the DDP gradient all-reduce (reference core/engine.py:455-468).
Two ways to get it, both covered by world-size-2 gloo tests (tests/test_dist_cpu.py):
  * `GradReducer` (below; `install_grad_reducer(model)`): the native exchange over the flat gradient arena, chunked and
    overlapped with the backward plan - what bench.py --mode train --gpus N times;
  * the reference's own `torch.nn.parallel.DistributedDataParallel(model)` wrapper: its gradient-ready hooks fire because
    train_engine._TrainStepFn delivers a (zero) gradient to every parameter through autograd when a process group is up and
    no GradReducer is installed; DDP then averages the arena views in place.
"""
import os

import torch


class Replicas:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.dist = None
        # Y6_FORCE_DIST=1: bring the process group up for a one-rank job too, so that a 1-GPU box runs the very code an
        # 8-GPU launch runs (RCCL communicator, barriers, the MAX reduce, the chunked gradient all-reduce)
        if self.world > 1 or os.environ.get("Y6_FORCE_DIST", "0") == "1":
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")   # container hostnames may not resolve
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            self.backend = backend
            if backend == "nccl":
                # bind this process to its GPU BEFORE the communicator exists: RCCL's lazy init and every
                # device-less barrier use the current device
                torch.cuda.set_device(self.local_rank)
            if not dist.is_initialized():
                dist.init_process_group(backend, init_method="env://", rank=self.rank, world_size=self.world)
            self.dist = dist

    @property
    def is_main(self):
        return self.rank == 0

    def device(self):
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
            return torch.device("cuda", self.local_rank)
        return torch.device("cpu")

    def barrier(self):
        if self.dist is not None:
            if self.backend == "nccl":
                self.dist.barrier(device_ids=[self.local_rank])
            else:
                self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        """MAX all-reduce of a host scalar (the bench reports the slowest rank's time)."""
        if self.dist is None:
            return float(value)
        dev = torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def shard(self, n_items: int) -> range:
        """Contiguous, balanced shard of range(n_items) for this rank (images of an eval set)."""
        base, rem = divmod(n_items, self.world)
        start = self.rank * base + min(self.rank, rem)
        return range(start, start + base + (1 if self.rank < rem else 0))

    def throughput(self, units_per_rank: int, steps: int, elapsed_max: float) -> float:
        """Whole-job units/s: every rank processed units_per_rank x steps in (max over ranks) seconds."""
        return self.world * units_per_rank * steps / elapsed_max

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.barrier()
            self.dist.destroy_process_group()
            self.dist = None


class GradReducer:
    """The DDP gradient exchange of the training step (reference tools/train.py:120-125, core/engine.py:455-468:
    `DistributedDataParallel(model)`; 80.6 MB of fp32 gradients per step for YOLOv6-S), MI355X-style:

    * the gradients already live in ONE flat fp32 arena (train_engine.ParamArena), laid out in the order the backward
      pass finishes them - no bucket copy-in / copy-out as in torch DDP;
    * the arena is cut into a few large chunks (xGMI is point-to-point, 7 links x ~153 GB/s per GPU: ring all-reduce is
      per-link bound, so FEW LARGE messages beat many 25 MB buckets); chunk k is all-reduced (RCCL sum) on a side stream as
      soon as the backward ops that write it have been queued, overlapping the rest of the backward plan;
    * the average (1 / world size) is folded into the optimizer kernel (FusedSGD.step(grad_mul=...)), or applied here
      when `average=True`.
    BatchNorm statistics stay local (no SyncBN in the reference); buffers are not broadcast (rank 0's running statistics
    are what checkpoints keep, exactly as with DDP's rank-0 broadcast)."""

    def __init__(self, arena=None, bwd_marks=None, n_bwd_ops=None, replicas: "Replicas" = None, chunks=4, average=False):
        self.rep, self.average, self.chunks = replicas, average, chunks
        self.arena, self.segments, self.comm_stream = None, [], None
        if arena is not None:
            self.bind(arena, bwd_marks, n_bwd_ops)

    def bind(self, arena, bwd_marks, n_bwd_ops):
        """(Re)derive the segments for a training graph: which backward ops finalise which chunk of the arena.  Called
        lazily from run_backward when the model's graph was rebuilt (new input shape, `.half()`, re-parameterisation)."""
        chunks = self.chunks
        self.arena = arena
        self.n_bwd_ops = n_bwd_ops
        final_op = {}
        for op_end, params in bwd_marks:
            for p in params:
                final_op[id(p)] = max(final_op.get(id(p), 0), op_end)
        # chunk boundaries on parameter boundaries, ~equal element counts
        target = max(1, arena.numel // max(1, chunks))
        self.segments = []          # (first_op, last_op, lo, hi)
        lo, ready, prev_op = 0, 0, 0
        for p, off in zip(arena.params, arena.offsets):
            ready = max(ready, final_op.get(id(p), n_bwd_ops))
            hi = off + ((p.numel() + 3) // 4) * 4
            if hi - lo >= target or p is arena.params[-1]:
                end = max(ready, prev_op)
                if p is arena.params[-1]:
                    end, hi = n_bwd_ops, arena.numel
                self.segments.append((prev_op, end, lo, hi))
                prev_op, lo = end, hi

    def reduce_range(self, lo, hi, stream_ctx=None):
        if self.rep.dist is None:
            return
        t = self.arena.grad[lo:hi]
        self.rep.dist.all_reduce(t, op=self.rep.dist.ReduceOp.SUM)
        if self.average:
            t.mul_(1.0 / self.rep.world)

    def run_backward(self, graph, grads):
        """Segmented backward: launch the ops of segment k, then hand its arena chunk to RCCL on the side stream."""
        if graph.arena is not self.arena:
            self.bind(graph.arena, graph.bwd_marks, graph.n_bwd_ops)
        cuda = self.arena.grad.is_cuda
        if cuda and self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream()
        for first, last, lo, hi in self.segments:
            if last > first or first == 0:
                graph.backward(grads if first == 0 else None, first=first, last=last)
            if self.rep.dist is None:
                continue
            # the chunk's all-reduce waits for an event recorded on the launch stream: everything that writes the chunk - also
            # the weight-gradient ops the backward plan runs on its side stream - must have been ordered into that stream
            pending = getattr(getattr(graph, "bwd_plan", None), "side_pending", None)
            if pending is not None and pending():
                raise RuntimeError("yolov6_amd: the backward plan's side stream has not joined the launch stream: the gradient "
                                   "chunk would be all-reduced before its weight gradients are written")
            if cuda:
                ev = torch.cuda.Event()
                ev.record()
                self.comm_stream.wait_event(ev)
                with torch.cuda.stream(self.comm_stream):
                    self.reduce_range(lo, hi)
            else:
                self.reduce_range(lo, hi)
        if cuda and self.rep.dist is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def install(self, model):
        model.__dict__["_y6_backward_hook"] = self.run_backward


def install_grad_reducer(model, replicas: "Replicas" = None, chunks=4, average=True):
    """The native gradient exchange instead of a `DistributedDataParallel` wrapper: after this, `loss.backward()` on every
    rank leaves the all-reduced (by default: averaged, DDP's convention) gradient in `p.grad`.  A no-op reducer when the
    job has one rank.  `model` is the bare module (`de_parallel(model)`), not a DDP wrapper."""
    red = GradReducer(replicas=replicas or Replicas(), chunks=chunks, average=average)
    red.install(model)
    return red
