"""Network plug-in with the reference's interface (``network/um_v1.py:16,71`` of melonwan/denseReg).

The reference loads this module by name (``--net_module um_v1``) through ``importlib``
(``model/hourglass_um_crop_tiny.py:863-867``) and calls

    detect_net(dm_inputs, cfgs, coms, num_jnt, is_training=True, scope='') -> end_points

with ``end_points = {'hm_outs': [...], 'hm3_outs': [...], 'um_outs': [...]}``, one NHWC tensor per stack
(``um_v1.py:72-75,170-172,185``).  Here the graph is not built in Python: the call runs the HIP engine
(``libdensereg_hip.so``) and hands back device tensors.  ``cfgs``/``coms`` are accepted and unused inside
the net, exactly like the reference.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import flags
from ..engine import Engine

TOWER_NAME = 'um_v1'          # um_v1.py:16 (part of the cache-dir name, hourglass_um_crop_tiny.py:534-535)

_engines: Dict[tuple, Engine] = {}


def get_engine(num_jnt: int, in_hw: int, max_batch: int, device: int, training: bool) -> Engine:
    """One engine per (config, device); flags are read when the engine is first needed, like the
    reference reads ``FLAGS.num_stack/num_fea/kernel_size`` at graph-build time (um_v1.py:40,56,93,124)."""
    F = flags.FLAGS
    precision = getattr(F, 'precision', 'f32')
    key = (F.num_stack, F.num_fea, num_jnt, in_hw, F.kernel_size, device, training, precision)
    eng = _engines.get(key)
    if eng is None or eng.max_batch < max_batch:
        eng = Engine(F.num_stack, F.num_fea, num_jnt, in_hw, F.kernel_size, max_batch, device, training)
        if precision != 'f32':
            eng.set_precision(precision)                 # before any load_params: the packed weights follow the precision
        _engines[key] = eng
    return eng


def detect_net(dm_inputs: torch.Tensor, cfgs, coms, num_jnt: int, is_training: bool = True, scope: str = '',
               engine: Optional[Engine] = None, dropout_seed: int = 0) -> Dict[str, List[torch.Tensor]]:
    B, H, W, _ = dm_inputs.shape
    if H != W or H not in (128, 256, 512):
        raise ValueError('unknown input depth map shape')            # um_v1.py:106-107
    eng = engine or get_engine(num_jnt, H, B, dm_inputs.device.index or 0, is_training)
    end_points = {'hm_outs': [], 'hm3_outs': [], 'um_outs': []}
    if is_training:
        eng.forward_train(dm_inputs, seed=dropout_seed)
    else:
        eng.forward_eval(dm_inputs, want_maps=False)
    for s in range(eng.num_stack):
        hm, hm3, um = eng.read_maps(B, s)
        end_points['hm_outs'].append(hm)
        end_points['hm3_outs'].append(hm3)
        end_points['um_outs'].append(um)
    return end_points
