"""One-process-per-GPU replica harness (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" on CPU).

The hot path is a strictly sequential autoregressive chain per image (257 LLM steps, each with a
16-step ODE), so it does not shard within one generation; independent prompts/images are the
natural unit and ranks run as REPLICAS with no data-path collective.  The only
communication is the timing harness: a barrier on both sides of the timed region and a MAX
reduction of the per-rank wall time.
"""
import os
import time

import torch


class ReplicaGroup:
    def __init__(self, backend=None, device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            if backend is None:
                backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = torch.device(device)
            if not dist.is_initialized():
                dist.init_process_group(backend, **kw)
            self.dist = dist

    def seed(self, base):
        """Distinct data per replica (weak scaling: every rank does the same AMOUNT of different work)."""
        return base + self.rank

    def barrier(self):
        cuda = self.device is not None and torch.device(self.device).type == "cuda"
        if cuda:
            torch.cuda.synchronize()          # this rank's GPU work is done before it reports to the barrier
        if self.dist is not None:
            self.dist.barrier()
        if cuda:
            torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn, steps, step_ms=None):
        """barrier -> `steps` calls of fn() -> barrier; returns MAX over ranks of the wall time.  step_ms (a list): filled with THIS
        rank's per-step milliseconds, from events recorded on the current stream between the steps (no extra host sync)."""
        cuda = step_ms is not None and self.device is not None and torch.device(self.device).type == "cuda"
        self.barrier()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if cuda else None
        t0 = time.perf_counter()
        out = None
        if cuda:
            ev[0].record()
        for i in range(steps):
            out = fn()
            if cuda:
                ev[i + 1].record()
        self.barrier()
        dt = time.perf_counter() - t0
        if cuda:
            step_ms.extend(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
        return self.max_over_ranks(dt), out

    def total(self, per_rank_units):
        """Whole-job units processed (all ranks do the same amount)."""
        return per_rank_units * self.world

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
