"""Multi-GPU plumbing for the hot path: one process per GPU, independent replicas.

Inference shards by independent images: every rank owns its own batch and its own
native plan; there is NO data-path collective.  The process group (backend "nccl" = RCCL on ROCm,
"gloo" on CPU) is used only for barriers and one MAX all-reduce of the elapsed time, as the bench
contract requires.  (The DDP gradient all-reduce of the training step - reference
core/engine.py:463-466 - belongs to the training row and is not part of this module yet.)
"""
import os

import torch


class Replicas:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.dist = None
        if self.world > 1:
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
