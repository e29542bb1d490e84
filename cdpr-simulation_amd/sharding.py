"""Multi-GPU placement: robots are independent (the reference simulates exactly one, PLG.cpp:202-246 touches one
model's joints), so a batch shards by contiguous blocks with NO collective on the data path.  One process per GPU;
`torch.distributed` is used only as the rendezvous (barrier, max of elapsed time) that bench.py's contract asks for.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Tuple


def shard_range(rank: int, world: int, total: int) -> Tuple[int, int]:
    """Robots [lo, hi) owned by `rank`: contiguous blocks whose sizes differ by at most one."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@dataclass
class RankContext:
    rank: int = 0
    local_rank: int = 0
    world: int = 1
    _dist: object = None

    @classmethod
    def from_env(cls, backend: str = "nccl") -> "RankContext":
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world == 1:
            return cls(rank, local_rank, world, None)
        import torch
        import torch.distributed as dist

        if backend == "nccl" and torch.cuda.is_available() and torch.cuda.device_count() > local_rank:
            torch.cuda.set_device(local_rank)  # "nccl" is RCCL on ROCm
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:  # CPU-only box, or more ranks than GPUs: the rendezvous does not need the GPU
            dist.init_process_group(backend="gloo")
        return cls(rank, local_rank, world, dist)

    def barrier(self) -> None:
        if self._dist is not None:
            import torch

            if self._dist.get_backend() == "nccl":
                torch.cuda.synchronize()
            self._dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self._dist is None:
            return value
        import torch

        dev = "cuda" if self._dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def close(self) -> None:
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None
