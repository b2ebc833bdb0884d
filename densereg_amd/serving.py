"""Inference replicas on ONE GPU (north-star: "inference = replicas", no collective on the path).

``test_model.test`` (reference ``model/test_model.py:14-94``) pushes one batch after another through forward(eval) + vote; the
batches are independent.  A single stream of these kernels leaves the chip idle at every launch boundary and through the
small-grid hourglass levels (~150 launches per batch), so ``ReplicaPool`` keeps k engines -- k handles with the same weights,
each on its own stream -- and hands consecutive batches to them in turn: the kernels of batch i+1 fill the gaps of batch i
(measured on MI355X, ICVL S=2 F=128 B=40: 7866 crops/s with one replica, 8961 with two, 9421 with three;
``profiles/r03_experiments.md``).  Results are delivered in submission order; each ``submit`` is ordered after whatever the
caller's stream held at the time, and ``wait`` orders the caller's stream behind a batch's result.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .engine import Engine


class ReplicaPool:
    def __init__(self, replicas: int, num_stack=2, num_fea=128, num_jnt=16, in_hw=128, kernel_size=3, max_batch=40, device: int = 0):
        assert replicas >= 1
        self.device = torch.device('cuda', device)
        self.engines: List[Engine] = [Engine(num_stack, num_fea, num_jnt, in_hw, kernel_size, max_batch, device, training=False)
                                      for _ in range(replicas)]
        # high-priority streams: the runtime maps streams onto a few hardware queues per priority level, least used first -- on
        # their own level the replicas do not end up sharing a queue with each other or with the caller's streams (measured: two
        # replicas on normal-priority streams beside a training engine's streams ran at 6543 crops/s, slower than one replica)
        prio = torch.cuda.Stream.priority_range()[1] if hasattr(torch.cuda.Stream, 'priority_range') else -1
        self.streams = [torch.cuda.Stream(self.device, priority=prio) for _ in range(replicas)]
        self.num_jnt = num_jnt
        self._next = 0

    def __len__(self):
        return len(self.engines)

    def set_precision(self, precision: str):
        for e in self.engines:
            e.set_precision(precision)

    def load_params(self, params: Dict[str, np.ndarray]):
        for e in self.engines:
            e.load_params(params)

    def param_infos(self):
        return self.engines[0].param_infos()

    def conv_flops_per_crop(self) -> float:
        return self.engines[0].conv_flops_per_crop()

    def norm_dm(self, dm_mm, com):
        return self.engines[0].norm_dm(dm_mm, com)

    def submit(self, dm_norm: torch.Tensor, cfg: torch.Tensor, com: torch.Tensor, out: Optional[torch.Tensor] = None):
        """forward(eval) + vote of one batch on the next replica; returns (xyz, ticket).  ``xyz`` is valid for the caller's stream
        after ``wait(ticket)`` (or a device synchronisation)."""
        i = self._next
        self._next = (i + 1) % len(self.engines)
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream(self.device))          # the batch's inputs (and `out`) as the caller left them
        with torch.cuda.stream(s):
            xyz = self.engines[i].infer(dm_norm, cfg, com, out=out)
            ev = torch.cuda.Event()
            ev.record(s)
        for t in (dm_norm, cfg, com, xyz):
            t.record_stream(s)                                        # the caching allocator must not recycle them under the replica
        return xyz, ev

    def wait(self, ticket):
        torch.cuda.current_stream(self.device).wait_event(ticket)

    def infer(self, dm_norm, cfg, com, out=None):
        """Synchronous form (one batch, caller's stream order): what ``Engine.infer`` does, on the next replica."""
        xyz, ev = self.submit(dm_norm, cfg, com, out)
        self.wait(ev)
        return xyz

    def close(self):
        torch.cuda.synchronize(self.device)
        for e in self.engines:
            e.close()
