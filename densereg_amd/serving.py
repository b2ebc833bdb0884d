"""Inference replicas on ONE GPU (north-star: "inference = replicas", no collective on the path).

``test_model.test`` (reference ``model/test_model.py:14-94``) pushes one batch after another through forward(eval) + vote; the
batches are independent.  A single stream of these kernels leaves the chip idle at every launch boundary and through the
small-grid hourglass levels (~150 launches per batch), so ``ReplicaPool`` keeps k engines -- k handles with the same weights,
each on its own stream -- and hands consecutive batches to them in turn: the kernels of batch i+1 fill the gaps of batch i
(measured on MI355X, ICVL S=2 F=128 B=40: 7866 crops/s with one replica, 8961 with two, 9421 with three;
``profiles/r03_experiments.md``).  Results are delivered in submission order; each ``submit`` is ordered after whatever the
caller's stream held at the time, and ``wait`` orders the caller's stream behind a batch's result.

``merge`` > 1 adds the other half of a serving loop: consecutive submitted batches are staged side by side and run as ONE
forward(eval) + vote of ``merge`` x the rows -- crops are independent on this path (moving statistics, no batch coupling),
so the result of a crop does not depend on its neighbours, only the kernels' grids grow (a 2x2-pixel layer of 40 crops is
160 rows of work for 256 CUs).  Measured on MI355X, ICVL S=2 F=128, batches of 40: 3 replicas 9686 crops/s; 2 replicas
merging 3 batches 10141, merging 5 batches 10524; one replica merging 5: 9789.  A merged group is launched when its last
batch arrives, or by ``wait`` on one of its tickets / ``flush()``: latency grows by the wait for the group to fill.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .engine import Engine


class _Group:
    """merged batches of one launch on one replica (ReplicaPool, merge > 1); doubles as the ticket of its batches"""

    def __init__(self, replica: int, xyz_all: torch.Tensor):
        self.replica, self.xyz_all = replica, xyz_all
        self.rows = 0
        self.parts = []            # (caller's out tensor or None, first row, rows)
        self.event = None          # recorded behind the launch; None while the group is still filling


class ReplicaPool:
    def __init__(self, replicas: int, num_stack=2, num_fea=128, num_jnt=16, in_hw=128, kernel_size=3, max_batch=40, device: int = 0,
                 merge: int = 1):
        assert replicas >= 1 and merge >= 1
        self.device = torch.device('cuda', device)
        self.merge, self.max_batch, self.in_hw = int(merge), int(max_batch), int(in_hw)
        self.engines: List[Engine] = [Engine(num_stack, num_fea, num_jnt, in_hw, kernel_size, max_batch * self.merge, device, training=False)
                                      for _ in range(replicas)]
        # high-priority streams: the runtime maps streams onto a few hardware queues per priority level, least used first -- on
        # their own level the replicas do not end up sharing a queue with each other or with the caller's streams (measured: two
        # replicas on normal-priority streams beside a training engine's streams ran at 6543 crops/s, slower than one replica)
        prio = torch.cuda.Stream.priority_range()[1] if hasattr(torch.cuda.Stream, 'priority_range') else -1
        self.streams = [torch.cuda.Stream(self.device, priority=prio) for _ in range(replicas)]
        self.num_jnt = num_jnt
        self._next = 0
        self._open: Optional[_Group] = None          # the group being filled (merge > 1)
        self._stage = None
        if self.merge > 1:                           # per replica: where the batches of a group are put side by side
            rows = max_batch * self.merge
            self._stage = [(e.new(rows, in_hw, in_hw), e.new(rows, 6), e.new(rows, 3)) for e in self.engines]

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
        after ``wait(ticket)`` (or ``flush()`` and a device synchronisation)."""
        if self.merge > 1:
            return self._submit_merged(dm_norm, cfg, com, out)
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

    def _submit_merged(self, dm_norm, cfg, com, out):
        b = dm_norm.shape[0]
        if b > self.max_batch:
            raise ValueError('batch of %d crops, the pool was built for %d' % (b, self.max_batch))
        g = self._open
        if g is None:
            i = self._next
            self._next = (i + 1) % len(self.engines)
            g = self._open = _Group(i, self.engines[i].new(self.max_batch * self.merge, 3 * self.num_jnt))
        i = g.replica
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream(self.device))
        sdm, scfg, scom = self._stage[i]
        r0 = g.rows
        with torch.cuda.stream(s):                                     # (stream order: behind the previous group's launch on this replica)
            sdm[r0:r0 + b].copy_(dm_norm.reshape(b, self.in_hw, self.in_hw))
            scfg[r0:r0 + b].copy_(cfg)
            scom[r0:r0 + b].copy_(com)
        for t in (dm_norm, cfg, com) + ((out,) if out is not None else ()):
            t.record_stream(s)
        g.parts.append((out, r0, b))
        g.rows += b
        xyz = out if out is not None else g.xyz_all[r0:r0 + b]
        if len(g.parts) == self.merge:
            self._launch(g)
        return xyz, g

    def _launch(self, g: _Group):
        i = g.replica
        s = self.streams[i]
        sdm, scfg, scom = self._stage[i]
        with torch.cuda.stream(s):
            self.engines[i].infer(sdm[:g.rows], scfg[:g.rows], scom[:g.rows], out=g.xyz_all[:g.rows])
            for out, r0, b in g.parts:
                if out is not None:
                    out.copy_(g.xyz_all[r0:r0 + b].reshape(out.shape))
            g.event = torch.cuda.Event()
            g.event.record(s)
        g.xyz_all.record_stream(s)
        if self._open is g:
            self._open = None

    def flush(self):
        """launch the group that is still filling (merge > 1); nothing to do otherwise"""
        if self._open is not None and self._open.rows > 0:
            self._launch(self._open)

    def wait(self, ticket):
        if isinstance(ticket, _Group):
            if ticket.event is None:
                self._launch(ticket)
            ticket = ticket.event
        torch.cuda.current_stream(self.device).wait_event(ticket)

    def infer(self, dm_norm, cfg, com, out=None):
        """Synchronous form (one batch, caller's stream order): what ``Engine.infer`` does, on the next replica."""
        xyz, ev = self.submit(dm_norm, cfg, com, out)
        self.wait(ev)
        return xyz

    def close(self):
        self.flush()
        torch.cuda.synchronize(self.device)
        for e in self.engines:
            e.close()
