"""PyTorch-CPU restatement of the reference network forward / loss / gradients.

TEST INFRASTRUCTURE -- parity unpinned (see oracle/__init__.py).  Follows
``network/um_v1.py`` via ``oracle.graph.walk_detect_net`` and the layer wrappers
of ``network/slim/ops.py``:

* ``conv``      -> ops.py:219-299  (tf.nn.conv2d NHWC x HWIO, SAME padding with the
                   extra pad on bottom/right, BatchReNorm xor bias, then ReLU)
* ``_batch_renorm`` -> ops.py:130-180 (biased moments, eps inside the sqrt, r/d
                   stop-gradient; eval: gamma*(x-mu_mov)*rsqrt(var_mov+eps)+beta)
* ``max_pool``  -> ops.py:640-669 (padding never wins the max)
* ``upsample2`` -> ops.py:671-677 (legacy nearest: src = floor(dst/2))
* ``tiny_dm``   -> um_v1.py:111   (bicubic, scale exactly 4, no half-pixel offset:
                   taps (0,1,0,0) => dm[:, ::4, ::4])

All public tensors are NHWC like the reference; torch NCHW is internal.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .graph import NetConfig, OpsBase, conv_specs, param_specs, same_pad, walk_detect_net

BN_DECAY = 0.99      # um_v1.py:9
BN_EPS = 0.001       # um_v1.py:10


def init_params(cfg: NetConfig, seed: int = 7, reference_init: bool = False) -> Dict[str, np.ndarray]:
    """Random parameters keyed by TF variable name.

    ``reference_init=True`` reproduces the reference initialisers (trunc-normal
    sigma=0.01 weights, zero bias/beta, unit gamma, moving stats (0,1); ops.py:86-128,272,290).
    The default is He-style weights and randomised BN state:
    with sigma=0.01 and identity BN the maps collapse to ~0 and the vote degenerates.
    """
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape, _ in param_specs(cfg):
        leaf = name.rsplit('/', 1)[1]
        if leaf == 'weights':
            if reference_init:
                w = torch.empty(shape).normal_(0, 1, generator=g)
                # resample outside 2 sigma (truncated normal)
                for _ in range(8):
                    bad = w.abs() > 2
                    if not bad.any():
                        break
                    w[bad] = torch.empty(int(bad.sum())).normal_(0, 1, generator=g)
                w = w * 0.01
            else:
                fan_in = shape[0] * shape[1] * shape[2]
                w = torch.empty(shape).normal_(0, 1, generator=g) * math.sqrt(2.0 / fan_in)
            out[name] = w.numpy().astype(np.float32)
        elif leaf == 'biases':
            v = torch.zeros(shape) if reference_init else torch.empty(shape).normal_(0, 0.05, generator=g)
            out[name] = v.numpy().astype(np.float32)
        elif leaf == 'beta':
            v = torch.zeros(shape) if reference_init else torch.empty(shape).normal_(0, 0.1, generator=g)
            out[name] = v.numpy().astype(np.float32)
        elif leaf == 'gamma':
            v = torch.ones(shape) if reference_init else torch.empty(shape).uniform_(0.5, 1.5, generator=g)
            out[name] = v.numpy().astype(np.float32)
        elif leaf == 'moving_mean':
            v = torch.zeros(shape) if reference_init else torch.empty(shape).normal_(0, 0.1, generator=g)
            out[name] = v.numpy().astype(np.float32)
        elif leaf == 'moving_variance':
            v = torch.ones(shape) if reference_init else torch.empty(shape).uniform_(0.5, 1.5, generator=g)
            out[name] = v.numpy().astype(np.float32)
        elif leaf == 'r_max':
            out[name] = np.ones(shape, np.float32)
        elif leaf in ('d_max', 'curr_t'):
            out[name] = np.zeros(shape, np.float32)
        else:
            raise KeyError(name)
    return out


class _ConvBf16Operands(torch.autograd.Function):
    """A convolution whose matrix-core operands are bfloat16 in all three products (include/densereg.h,
    dr_set_precision): forward r(x) * r(w), input gradient r(gy) * r(w), weight gradient r(x) * r(gy), r = round to
    nearest-even bfloat16, products and sums in the tensors' own precision.  The rounding itself has no gradient."""

    @staticmethod
    def forward(ctx, x, w, stride):
        r = lambda t: t.to(torch.bfloat16).to(t.dtype)
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return F.conv2d(r(x), r(w), stride=stride)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        r = lambda t: t.to(torch.bfloat16).to(t.dtype)
        gx = torch.nn.grad.conv2d_input(x.shape, r(w), r(gy), stride=ctx.stride) if ctx.needs_input_grad[0] else None
        gw = torch.nn.grad.conv2d_weight(r(x), w.shape, r(gy), stride=ctx.stride) if ctx.needs_input_grad[1] else None
        return gx, gw, None


class TorchOps(OpsBase):
    """Computing backend for ``walk_detect_net``.  Tensors are NCHW torch tensors."""

    def __init__(self, cfg: NetConfig, params: Dict[str, torch.Tensor], is_training: bool,
                 dropout_masks: Optional[List[torch.Tensor]] = None, dtype=torch.float32,
                 record: Optional[dict] = None, conv_operands: str = 'f32', switches: Optional[dict] = None):
        super().__init__(cfg)
        # ``switches`` (tests only): the DISCRETE decisions of another evaluation of the same graph, injected so that two evaluations in
        # different precisions take the same branches and what is left between their gradients is arithmetic alone --
        #   switches['relu'](conv name)  -> NHWC bool array: which units of that conv's ReLU are open (replaces torch.relu)
        #   switches['act'](conv name)   -> NHWC array of what the other evaluation stored for that conv (after its residual add): a
        #                                   max-pool takes its argmax positions from THAT tensor (first maximum in scan order)
        self.switches = switches
        # 'bf16': what a bf16 matrix-core path computes (BASELINE config 5; include/densereg.h dr_set_precision) --
        # both operands of every k != 7 convolution rounded to bfloat16 (nearest even), products and sums in fp32
        self.conv_operands = conv_operands
        self.p = params
        self.is_training = is_training
        self.dropout_masks = dropout_masks
        self._drop_i = 0
        self.dtype = dtype
        self.bn_updates: Dict[str, torch.Tensor] = {}
        self.record = record        # name -> NHWC numpy of every conv output (post activation)
        self._producer = {}         # id(tensor) -> (conv name, tensor) for '+res' records
        self._stored = {}           # id(tensor) -> (conv name, tensor): what the engine stores under that conv's name (its output, or the
                                    # residual sum fused behind it) -- the max-pools' inputs, for ``switches``

    def channels(self, x):
        return x.shape[1]

    # -- ops.py:43-185 ---------------------------------------------------------------
    def _batch_renorm(self, x, scope):
        b = scope + '/BatchReNorm/'
        beta, gamma = self.p[b + 'beta'], self.p[b + 'gamma']
        mm, mv = self.p[b + 'moving_mean'].detach(), self.p[b + 'moving_variance'].detach()
        sh = (1, -1, 1, 1)
        if not self.is_training:
            inv = torch.rsqrt(mv + BN_EPS) * gamma           # tf.nn.batch_normalization
            return x * inv.view(sh) + (beta - mm * inv).view(sh)
        r_max = self.p[b + 'r_max'].detach()
        d_max = self.p[b + 'd_max'].detach()
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean.view(sh)) ** 2).mean(dim=(0, 2, 3))     # biased (tf.nn.moments)
        std = torch.sqrt(var + BN_EPS)
        mstd = torch.sqrt(mv + BN_EPS)
        # "read old, then update" (SURVEY Appendix C.2): r/d use the values before this step
        r = torch.maximum(torch.minimum(std / mstd, r_max), 1.0 / r_max).detach()
        d = torch.maximum(torch.minimum((mean - mm) / mstd, d_max), -d_max).detach()
        y = (x - mean.view(sh)) * torch.rsqrt(var + BN_EPS).view(sh)
        y = y * r.view(sh) + d.view(sh)
        y = y * gamma.view(sh) + beta.view(sh)
        self.bn_updates[scope] = {'mean': mean.detach(), 'var': var.detach(), 'r': r, 'd': d}
        return y

    # -- ops.py:219-299 --------------------------------------------------------------
    def conv(self, x, cout, k, stride, bn, relu, wd):
        name = self._namer.next_conv()
        w = self.p[name + '/weights']                      # HWIO
        assert tuple(w.shape) == (k, k, x.shape[1], cout), (name, tuple(w.shape), x.shape)
        H, W = x.shape[2], x.shape[3]
        pt, pb = same_pad(H, k, stride)
        pl, pr = same_pad(W, k, stride)
        xp = F.pad(x, (pl, pr, pt, pb)) if (pt or pb or pl or pr) else x
        if self.conv_operands == 'bf16' and k != 7:          # the 1-channel stem stays on the fp32 direct kernel
            y = _ConvBf16Operands.apply(xp, w.permute(3, 2, 0, 1), stride)
        else:
            y = F.conv2d(xp, w.permute(3, 2, 0, 1), stride=stride)
        if bn:
            y = self._batch_renorm(y, name)
        else:
            y = y + self.p[name + '/biases'].view(1, -1, 1, 1)
        if relu:
            if self.switches is not None:
                open_ = torch.from_numpy(np.ascontiguousarray(self.switches['relu'](name))).permute(0, 3, 1, 2)
                y = y * open_.to(y.dtype)
            else:
                y = torch.relu(y)
        if self.record is not None:
            self.record[name] = y.detach().permute(0, 2, 3, 1).contiguous().numpy()
        if self.record is not None or self.switches is not None:
            self._producer[id(y)] = (name, y)
            self._stored[id(y)] = (name, y)
        return y

    def add(self, a, b):
        out = a + b
        if self.record is not None or self.switches is not None:
            # the HIP engine fuses the residual add into the producing conv's epilogue: also record
            # '<conv>+res' so tests can compare what the engine stores for that conv.
            for t in (a, b):
                ent = self._producer.get(id(t))
                if ent is not None and ent[1] is t:
                    if self.record is not None:
                        self.record[ent[0] + '+res'] = out.detach().permute(0, 2, 3, 1).contiguous().numpy()
                    self._stored[id(out)] = (ent[0], out)            # (the conv whose stored output this sum is)
                    break
        return out

    def max_pool(self, x, k, s):
        H, W = x.shape[2], x.shape[3]
        pt, pb = same_pad(H, k, s)
        pl, pr = same_pad(W, k, s)
        pad = lambda t: F.pad(t, (pl, pr, pt, pb), value=float('-inf')) if (pt or pb or pl or pr) else t
        if self.switches is not None:
            # the other evaluation's choice of the maximum of every window (first maximum in scan order, as torch reports it), applied
            # to THIS evaluation's values
            ent = self._stored.get(id(x))
            assert ent is not None and ent[1] is x, 'max_pool input is not a stored conv output'
            other = torch.from_numpy(np.ascontiguousarray(self.switches['act'](ent[0]))).permute(0, 3, 1, 2)
            _, idx = F.max_pool2d(pad(other), k, s, return_indices=True)
            xp = pad(x)
            return xp.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
        return F.max_pool2d(pad(x), k, s)

    def upsample2(self, x):
        return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)

    def tiny_dm(self, dm):
        return dm[:, :, ::4, ::4]

    def uvd(self, tiny):
        B, _, h, w = tiny.shape
        # um_v1.py:113-116: meshgrid 'xy' => uu[i,j] = j/(w/2)-1, vv[i,j] = i/(h/2)-1
        jj = torch.arange(w, dtype=self.dtype)
        ii = torch.arange(h, dtype=self.dtype)
        uu = (jj / float(w / 2) - 1.0).view(1, 1, 1, w).expand(B, 1, h, w)
        vv = (ii / float(h / 2) - 1.0).view(1, 1, h, 1).expand(B, 1, h, w)
        return torch.cat([uu, vv, tiny], dim=1)

    def concat(self, xs):
        return torch.cat(xs, dim=1)

    def depth_mask(self, x, tiny):
        return torch.where(tiny < -0.9, torch.zeros_like(x), x)

    def dropout(self, x):
        # ops.py:710-728: training -> tf.nn.dropout(x, 0.5) (kept units scaled by 2)
        if not self.is_training:
            return x
        i = self._drop_i
        self._drop_i += 1
        if self.dropout_masks is None:
            return x                                    # keep_prob == 1 (parity tests)
        m = self.dropout_masks[i]                       # NHWC {0,1} keep mask
        return x * (m.permute(0, 3, 1, 2).to(self.dtype) * 2.0)


def to_torch_params(params: Dict[str, np.ndarray], dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in params.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
        leaf = k.rsplit('/', 1)[1]
        if requires_grad and leaf in ('weights', 'biases', 'beta', 'gamma'):
            t.requires_grad_(True)
        out[k] = t
    return out


def detect_net(cfg: NetConfig, tparams, dm_nhwc: torch.Tensor, is_training: bool,
               dropout_masks=None, record=None, conv_operands='f32', switches=None):
    """um_v1.detect_net: dm (B,H,W,1) normalised -> end_points with NHWC tensors + TorchOps."""
    dtype = dm_nhwc.dtype
    ops = TorchOps(cfg, tparams, is_training, dropout_masks, dtype, record, conv_operands, switches)
    hm, hm3, um = walk_detect_net(ops, dm_nhwc.permute(0, 3, 1, 2))
    nhwc = lambda t: t.permute(0, 2, 3, 1)
    return {'hm_outs': [nhwc(t) for t in hm], 'hm3_outs': [nhwc(t) for t in hm3],
            'um_outs': [nhwc(t) for t in um]}, ops


def forward_eval(cfg: NetConfig, params: Dict[str, np.ndarray], dm: np.ndarray, dtype=torch.float32,
                 record=None, conv_operands='f32'):
    """Eval-mode forward; returns dict of lists of NHWC numpy arrays."""
    with torch.no_grad():
        tp = to_torch_params(params, dtype)
        ep, _ = detect_net(cfg, tp, torch.from_numpy(dm).to(dtype), False, record=record, conv_operands=conv_operands)
    return {k: [t.contiguous().numpy() for t in v] for k, v in ep.items()}


def reg_loss(cfg: NetConfig, tparams) -> torch.Tensor:
    """losses.py:56-72: sum over convs with weight_decay>0 of wd * sum(w^2)/2."""
    tot = None
    for c in conv_specs(cfg):
        if c.weight_decay > 0:
            w = tparams[c.name + '/weights']
            t = c.weight_decay * (w * w).sum() * 0.5
            tot = t if tot is None else tot + t
    return tot


# ---- BatchReNorm state update (ops.py:134-153) --------------------------------------
def bn_state_update(params: Dict[str, np.ndarray], bn_updates, zero_debias: bool = True,
                    shadow: Optional[dict] = None):
    """Apply one micro-step of update ops to moving stats / r_max / d_max / curr_t in place.

    ``assign_moving_average(var, value, 0.99)``: [TF1.3-semantics] default zero_debias=True keeps
    shadow ``biased`` (init 0) and ``local_step`` and sets var = biased/(1-0.99^step).
    r_max <- 3/(1+2e^-t); d_max <- 5/((1+5/1e-3-1)e^-2t); t <- t+1e-5, all from the OLD t.
    """
    for scope, u in bn_updates.items():
        b = scope + '/BatchReNorm/'
        for key, val in (('moving_mean', u['mean']), ('moving_variance', u['var'])):
            val = val.numpy().astype(np.float64)
            if zero_debias:
                st = shadow.setdefault(b + key, {'biased': np.zeros_like(val), 'step': 0})
                st['biased'] = st['biased'] - (st['biased'] - val) * (1.0 - BN_DECAY)
                st['step'] += 1
                params[b + key] = (st['biased'] / (1.0 - BN_DECAY ** st['step'])).astype(np.float32)
            else:
                old = params[b + key].astype(np.float64)
                params[b + key] = (old - (old - val) * (1.0 - BN_DECAY)).astype(np.float32)
        t = float(params[b + 'curr_t'][0])
        params[b + 'r_max'] = np.array([3.0 / (1.0 + 2.0 * math.exp(-t))], np.float32)
        params[b + 'd_max'] = np.array([5.0 / ((1.0 + 5.0 / 1e-3 - 1.0) * math.exp(-2.0 * t))], np.float32)
        params[b + 'curr_t'] = np.array([t + 1e-5], np.float32)


def make_test_params(cfg: NetConfig, dm_norm: np.ndarray, seed: int = 7, rounds: int = 2) -> Dict[str, np.ndarray]:
    """"Trained-like" parameters for parity tests (last bullet).

    He-normal weights, then calibrated on ``dm_norm`` so that eval-mode activations stay O(1):
    moving stats := batch stats of a train-mode pass (then perturbed so r != 1, d != 0 in
    train-mode tests), head convs rescaled so hm/hm3/um land in a plausible range.  With the
    reference initialisers the maps collapse to ~0 and the vote degenerates to tie-breaking.
    """
    params = init_params(cfg, seed)
    specs = conv_specs(cfg)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    dm = torch.from_numpy(dm_norm).float()
    for it in range(rounds):
        rec = {}
        with torch.no_grad():
            _, ops = detect_net(cfg, to_torch_params(params), dm, True, None, record=rec)
        for scope, u in ops.bn_updates.items():
            b = scope + '/BatchReNorm/'
            params[b + 'moving_mean'] = u['mean'].numpy().astype(np.float32)
            params[b + 'moving_variance'] = u['var'].numpy().astype(np.float32)
        # rescale the linear heads (hm, hm3, um): bias convs without relu and cout in {J, 3J}
        for c in specs:
            if (not c.bn) and (not c.relu) and c.cout in (cfg.num_jnt, 3 * cfg.num_jnt):
                y = rec[c.name]
                s = float(y.std()) + 1e-6
                tgt_std, tgt_mean = (0.25, 0.15) if c.cout == cfg.num_jnt else (0.4, 0.0)
                params[c.name + '/weights'] = (params[c.name + '/weights'] * (tgt_std / s)).astype(np.float32)
                mu = y.mean(axis=(0, 1, 2)) - params[c.name + '/biases']
                params[c.name + '/biases'] = (tgt_mean - mu * (tgt_std / s)).astype(np.float32)
    for c in specs:
        if c.bn:
            b = c.name + '/BatchReNorm/'
            std = np.sqrt(params[b + 'moving_variance'] + BN_EPS)
            params[b + 'moving_mean'] = (params[b + 'moving_mean'] + 0.05 * std * rng.standard_normal(c.cout)).astype(np.float32)
            params[b + 'moving_variance'] = (params[b + 'moving_variance'] * rng.uniform(0.9, 1.1, c.cout)).astype(np.float32)
            # a "mid-training" renorm schedule point: t = 0.3 -> r_max ~1.21, d_max ~1.8e-3
            params[b + 'curr_t'] = np.array([0.3], np.float32)
            params[b + 'r_max'] = np.array([3.0 / (1.0 + 2.0 * math.exp(-0.3))], np.float32)
            params[b + 'd_max'] = np.array([5.0 / ((1.0 + 5.0 / 1e-3 - 1.0) * math.exp(-0.6))], np.float32)
    return params
