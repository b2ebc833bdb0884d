"""Structure of the reference network, backend-agnostic (TEST INFRASTRUCTURE).

``walk_detect_net`` restates the op sequence of ``network/um_v1.py:18-185`` against
an abstract ``ops`` backend; ``SpecOps`` (here) records shapes/names only and
``oracle.net.TorchOps`` computes.  Variable names reproduce TF default-name
uniquification in creation order (``network/slim/ops.py:266``
``variable_scope(scope, 'Conv')``; stem under ``hg_imgproc/``, ``um_v1.py:84``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass(frozen=True)
class NetConfig:
    """Flags the reference reads as process globals (hourglass_um_crop_tiny.py:55-60)."""
    num_stack: int = 2
    num_fea: int = 128
    num_jnt: int = 16
    in_hw: int = 128
    kernel_size: int = 3

    @property
    def out_hw(self) -> int:
        return self.in_hw // 4

    @property
    def num_resize(self) -> int:
        # um_v1.py:99-107
        return {512: 6, 256: 5, 128: 4}[self.in_hw]


@dataclass
class ConvSpec:
    name: str          # TF scope, e.g. 'hg_imgproc/Conv_3' or 'Conv_40'
    k: int
    stride: int
    cin: int
    cout: int
    bn: bool           # BatchReNorm (True) xor bias (False)  -- ops.py:284-296
    relu: bool
    weight_decay: float
    h_out: int
    w_out: int

    @property
    def flops_per_crop(self) -> int:
        return 2 * self.h_out * self.w_out * self.k * self.k * self.cin * self.cout


class _Namer:
    """TF variable_scope default-name uniquification ('Conv', 'Conv_1', ...) per parent scope."""

    def __init__(self):
        self.counts = {}
        self.prefix = ''

    def next_conv(self) -> str:
        n = self.counts.get(self.prefix, 0)
        self.counts[self.prefix] = n + 1
        base = 'Conv' if n == 0 else 'Conv_%d' % n
        return self.prefix + base


def walk_residual(ops, ins, num_out=None):
    """um_v1.py:18-48."""
    num_in = ops.channels(ins)
    if num_out is None:
        num_out = num_in
    half = num_in // 2
    k = ops.cfg.kernel_size
    out_1 = ops.conv(ins, half, 1, 1, bn=True, relu=True, wd=0.0005)
    out_1 = ops.conv(out_1, half, k, 1, bn=True, relu=True, wd=0.0005)
    out_1 = ops.conv(out_1, num_out, 1, 1, bn=True, relu=True, wd=0.0005)
    if num_out == num_in:
        out_2 = ins
    else:
        out_2 = ops.conv(ins, num_out, 1, 1, bn=True, relu=True, wd=0.0005)
    return ops.add(out_1, out_2)


def walk_hourglass(ops, ins, n):
    """um_v1.py:51-69."""
    upper1 = walk_residual(ops, ins)
    k = ops.cfg.kernel_size
    lower1 = ops.max_pool(ins, k, 2)
    lower1 = walk_residual(ops, lower1)
    if n > 1:
        lower2 = walk_hourglass(ops, lower1, n - 1)
    else:
        lower2 = lower1
    lower3 = walk_residual(ops, lower2)
    upper2 = ops.upsample2(lower3)
    return ops.add(upper1, upper2)


def walk_detect_net(ops, dm):
    """um_v1.py:71-185.  Returns (hm_outs, hm3_outs, um_outs)."""
    cfg = ops.cfg
    hm_outs, hm3_outs, um_outs = [], [], []
    ops.push_scope('hg_imgproc/')
    conv_1 = ops.conv(dm, 32, 7, 2, bn=True, relu=True, wd=0.0005)
    conv_2 = walk_residual(ops, conv_1, 64)
    pool_1 = ops.max_pool(conv_2, 2, 2)
    conv_3 = walk_residual(ops, pool_1)
    conv_4 = walk_residual(ops, conv_3, cfg.num_fea)
    hg_ins = conv_4
    ops.pop_scope()

    tiny_dm = ops.tiny_dm(dm)          # um_v1.py:111 (bicubic /4 == [::4, ::4])
    uvd = ops.uvd(tiny_dm)             # um_v1.py:113-121
    J = cfg.num_jnt
    for i in range(cfg.num_stack):
        hg_outs = walk_hourglass(ops, hg_ins, cfg.num_resize)
        ll = walk_residual(ops, hg_outs)
        ll = ops.conv(ll, cfg.num_fea, 1, 1, bn=True, relu=True, wd=0.0005)
        hm_out = ops.conv(ll, J, 1, 1, bn=False, relu=False, wd=0.0005)
        hm3_in = ops.concat([ll, uvd])
        hm3_in = walk_residual(ops, hm3_in, 128)
        hm3_out = ops.conv(hm3_in, J, 1, 1, bn=False, relu=False, wd=0.0005)

        um_in = ops.concat([hg_outs, hm_out, hm3_out])
        um_in = walk_residual(ops, walk_residual(ops, um_in, 256))
        um_in_mask = ops.concat([hg_outs, hm_out, hm3_out])
        um_in_mask = ops.depth_mask(um_in_mask, tiny_dm)     # um_v1.py:147-148
        um_in_mask = walk_residual(ops, walk_residual(ops, um_in_mask, 256))
        um_in_comb = ops.concat([um_in, um_in_mask])
        um_in_comb = walk_residual(ops, um_in_comb)
        um_in_comb = ops.concat([um_in_comb, uvd])
        um_full = ops.conv(um_in_comb, 512, 1, 1, bn=False, relu=True, wd=0.0005)
        um_full = ops.dropout(um_full)
        um_full = ops.conv(um_full, 512, 1, 1, bn=False, relu=True, wd=0.0005)
        um_full = ops.dropout(um_full)
        um_out = ops.conv(um_full, J * 3, 1, 1, bn=False, relu=False, wd=0.0005)
        hm_outs.append(hm_out)
        hm3_outs.append(hm3_out)
        um_outs.append(um_out)
        if i < cfg.num_stack - 1:
            tmp_out = ops.concat([hm_out, hm3_out, um_out])
            tmp_out_reshaped = ops.conv(tmp_out, cfg.num_fea, 1, 1, bn=False, relu=False, wd=0.0)
            inter = ops.conv(ll, cfg.num_fea, 1, 1, bn=False, relu=False, wd=0.0)
            hg_ins = ops.add(ops.add(hg_ins, tmp_out_reshaped), inter)
    return hm_outs, hm3_outs, um_outs


class OpsBase:
    def __init__(self, cfg: NetConfig):
        self.cfg = cfg
        self._namer = _Namer()

    def push_scope(self, s):
        self._namer.prefix = s

    def pop_scope(self):
        self._namer.prefix = ''


def same_out(h: int, s: int) -> int:
    return -(-h // s)


def same_pad(h: int, k: int, s: int) -> Tuple[int, int]:
    """TF 'SAME': total = max((ceil(h/s)-1)*s + k - h, 0); extra goes bottom/right."""
    total = max((same_out(h, s) - 1) * s + k - h, 0)
    lo = total // 2
    return lo, total - lo


class SpecOps(OpsBase):
    """Shape-only backend: tensors are (h, w, c) tuples."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.convs: List[ConvSpec] = []

    def channels(self, x):
        return x[2]

    def conv(self, x, cout, k, stride, bn, relu, wd):
        h, w, c = x
        ho, wo = same_out(h, stride), same_out(w, stride)
        self.convs.append(ConvSpec(self._namer.next_conv(), k, stride, c, cout, bn, relu, wd, ho, wo))
        return (ho, wo, cout)

    def add(self, a, b):
        assert a == b, (a, b)
        return a

    def max_pool(self, x, k, s):
        return (same_out(x[0], s), same_out(x[1], s), x[2])

    def upsample2(self, x):
        return (x[0] * 2, x[1] * 2, x[2])

    def tiny_dm(self, dm):
        return (dm[0] // 4, dm[1] // 4, 1)

    def uvd(self, tiny):
        return (tiny[0], tiny[1], 3)

    def concat(self, xs):
        assert all(x[:2] == xs[0][:2] for x in xs)
        return (xs[0][0], xs[0][1], sum(x[2] for x in xs))

    def depth_mask(self, x, tiny):
        return x

    def dropout(self, x):
        return x


def conv_specs(cfg: NetConfig) -> List[ConvSpec]:
    ops = SpecOps(cfg)
    walk_detect_net(ops, (cfg.in_hw, cfg.in_hw, 1))
    return ops.convs


def param_specs(cfg: NetConfig):
    """[(tf_variable_name, shape, trainable)] in TF creation order (ops.py:87-128, 276-295)."""
    out = []
    for c in conv_specs(cfg):
        out.append((c.name + '/weights', (c.k, c.k, c.cin, c.cout), True))
        if c.bn:
            b = c.name + '/BatchReNorm/'
            out.append((b + 'beta', (c.cout,), True))
            out.append((b + 'gamma', (c.cout,), True))
            out.append((b + 'moving_mean', (c.cout,), False))
            out.append((b + 'moving_variance', (c.cout,), False))
            out.append((b + 'r_max', (1,), False))
            out.append((b + 'd_max', (1,), False))
            out.append((b + 'curr_t', (1,), False))
        else:
            out.append((c.name + '/biases', (c.cout,), True))
    return out


def trainable_names(cfg: NetConfig):
    return [n for n, _, t in param_specs(cfg) if t]
