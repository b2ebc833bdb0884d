"""Shared test plumbing: one ``Backend`` interface over the C ABI for (a) the real HIP library on a
GPU and (b) the host-fiber emulation build of the same kernel sources (CPU, tests only)."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from densereg_amd import _lib  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


class Backend:
    """Uploads numpy arrays, calls the ABI, downloads results.  Subclasses define memory."""
    name = '?'

    def dev(self, a):               # numpy -> device buffer object
        raise NotImplementedError

    def empty(self, shape, dtype=np.float32):
        raise NotImplementedError

    def ptr(self, d):               # device buffer object -> address (or None)
        raise NotImplementedError

    def host(self, d):              # device buffer object -> numpy
        raise NotImplementedError

    def sync(self):
        pass

    stream = None

    # ---- helpers over the ABI -------------------------------------------------------------------
    def handle(self, cfg, max_batch, training=False):
        return _lib.Handle(self.lib, cfg.num_stack, cfg.num_fea, cfg.num_jnt, cfg.in_hw, cfg.kernel_size, max_batch, 0,
                           training)

    def conv2d(self, x, w, scale=None, shift=None, relu=False, res=None, rowmask=None, thresh=0.0, want_stats=False):
        B, H, W, Cin = x.shape
        k, _, _, Cout = w.shape
        x_cs = -(-Cin // 4) * 4
        y_cs = Cout + 3
        xp = np.full((B, H, W, x_cs), np.nan, np.float32)       # pad channels are poison: must not be read
        xp[..., :Cin] = x
        d_x, d_w = self.dev(xp), self.dev(np.ascontiguousarray(w, np.float32))
        d_y = self.dev(np.full((B, H, W, y_cs), -777.0, np.float32))
        d_sc = None if scale is None else self.dev(scale)
        d_sh = None if shift is None else self.dev(shift)
        d_res = None if res is None else self.dev(np.ascontiguousarray(res, np.float32))
        d_m = None if rowmask is None else self.dev(rowmask)
        d_st = self.dev(np.zeros((2, Cout), np.float64)) if want_stats else None
        rc = self.dbg.dr_dbg_conv2d(B, H, W, Cin, Cout, k, self.ptr(d_x), x_cs, self.ptr(d_w), self.ptr(d_sc),
                                    self.ptr(d_sh), int(relu), self.ptr(d_res), 0 if res is None else res.shape[-1],
                                    self.ptr(d_m), thresh, self.ptr(d_y), y_cs, self.ptr(d_st), self.stream)
        assert rc == 0, rc
        self.sync()
        yp = self.host(d_y)
        assert np.all(yp[..., Cout:] == -777.0), 'conv kernel wrote outside its channel range'
        y = yp[..., :Cout].copy()
        return (y, self.host(d_st)) if want_stats else y

    def wgrad(self, x, g, k, T, nsplit, rowmask=None, thresh=0.0):
        B, H, W, Cin = x.shape
        Cout = g.shape[-1]
        x_cs, g_cs = -(-Cin // 4) * 4 + 4, -(-Cout // 4) * 4
        xp = np.full((B, H, W, x_cs), np.nan, np.float32)       # pad channels are poison: must not reach the result
        xp[..., :Cin] = x
        gp = np.full((B, H, W, g_cs), np.nan, np.float32)
        gp[..., :Cout] = g
        d_x, d_g = self.dev(xp), self.dev(gp)
        d_m = None if rowmask is None else self.dev(rowmask)
        d_w = self.dev(np.full((k, k, Cin, Cout), -777.0, np.float32))
        rc = self.dbg.dr_dbg_wgrad(B, H, W, Cin, Cout, k, self.ptr(d_x), x_cs, self.ptr(d_g), g_cs, self.ptr(d_m), thresh,
                                   T, nsplit, self.ptr(d_w), self.stream)
        assert rc == 0, rc
        self.sync()
        return self.host(d_w)

    def forward_eval(self, h, ndm):
        B, J, m = ndm.shape[0], h.cfg.num_jnt, h.cfg.in_hw // 4
        d_dm = self.dev(ndm)
        hm, hm3, um = self.empty((B, m, m, J)), self.empty((B, m, m, J)), self.empty((B, m, m, 3 * J))
        h.call('dr_forward_eval', B, self.ptr(d_dm), self.ptr(hm), self.ptr(hm3), self.ptr(um), self.stream)
        self.sync()
        return self.host(hm), self.host(hm3), self.host(um)

    def vote(self, h, hm, hm3, um, ndm, cfg, com):
        B, J = ndm.shape[0], h.cfg.num_jnt
        xyz = self.empty((B, 3 * J))
        args = [self.dev(np.ascontiguousarray(a, np.float32)) for a in (hm, hm3, um, ndm, cfg, com)]
        h.call('dr_vote', B, *[self.ptr(a) for a in args], self.ptr(xyz), self.stream)
        self.sync()
        return self.host(xyz)

    def infer(self, h, ndm, cfg, com):
        B, J = ndm.shape[0], h.cfg.num_jnt
        xyz = self.empty((B, 3 * J))
        args = [self.dev(np.ascontiguousarray(a, np.float32)) for a in (ndm, cfg, com)]
        h.call('dr_infer', B, *[self.ptr(a) for a in args], self.ptr(xyz), self.stream)
        self.sync()
        return self.host(xyz)

    def norm_dm(self, h, dm, com):
        out = self.empty(dm.shape)
        d_dm, d_com = self.dev(dm), self.dev(com)
        h.call('dr_norm_dm', dm.shape[0], self.ptr(d_dm), self.ptr(d_com), self.ptr(out), self.stream)
        self.sync()
        return self.host(out)

    def read_activation(self, h, scope, shape):
        self.sync()
        a = np.empty(shape, np.float32)
        h.call('dr_read_activation', scope.encode(), shape[0], a.ctypes.data, a.size)
        return a


class EmuBackend(Backend):
    name = 'emu'

    def __init__(self):
        from tests.emu import load_emu
        self.lib = load_emu()
        self.dbg = self.lib                 # the emulator build carries the dr_dbg_* hooks itself
        # The emulator runs the fp32-MFMA kernels unless a test asks for the x3 ones (dr_dbg_force_x3): a fiber-emulated bf16 MFMA with
        # LDS transpose reads is several times slower than the emulated fp32 MFMA, and with the x3 weight gradients on by default the
        # CPU suite took 16 minutes instead of 10.  The x3 kernels keep their own emulator tests (test_conv_x3_..., test_wgrad_x3_...);
        # at network level they are covered on the GPU (by default and, once per round, on every layer: DR_CONV_X3=2).
        self.x3_default = 0
        self.dbg.dr_dbg_force_x3(self.x3_default)

    def dev(self, a):
        return np.ascontiguousarray(a).copy()

    def empty(self, shape, dtype=np.float32):
        return np.empty(shape, dtype)

    def ptr(self, d):
        return None if d is None else d.ctypes.data

    def host(self, d):
        return d


class GpuBackend(Backend):
    name = 'gpu'
    x3_default = -1                         # the product's rule (DR_CONV_X3 / DR_WGRAD_X3)

    def __init__(self):
        import torch
        self.torch = torch
        self.lib = _lib.load()              # the PRODUCT library: every handle of the GPU tests lives in it
        self.dbg = _lib.load_debug()        # same sources + dr_dbg_* hooks: single-kernel entry points only
        self.device = torch.device('cuda', 0)
        self.stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def empty(self, shape, dtype=np.float32):
        td = {np.float32: self.torch.float32, np.float64: self.torch.float64, np.uint8: self.torch.uint8}[dtype]
        return self.torch.empty(tuple(shape), dtype=td, device=self.device)

    def ptr(self, d):
        return None if d is None else d.data_ptr()

    def host(self, d):
        return d.cpu().numpy()

    def sync(self):
        self.torch.cuda.synchronize(self.device)


def ref_conv2d(x, w, scale=None, shift=None, relu=False, res=None, rowmask=None, thresh=0.0):
    """fp64 torch reference of one fused conv (returns fused output, raw conv)."""
    import torch
    import torch.nn.functional as F
    xt = torch.from_numpy(x).double()
    if rowmask is not None:
        m = torch.from_numpy(rowmask.reshape(x.shape[0], x.shape[1], x.shape[2], 1)).double()
        xt = torch.where(m < thresh, torch.zeros_like(xt), xt)
    k = w.shape[0]
    raw = F.conv2d(xt.permute(0, 3, 1, 2), torch.from_numpy(w).double().permute(3, 2, 0, 1), padding=k // 2).permute(0, 2, 3, 1)
    y = raw
    if scale is not None:
        y = y * torch.from_numpy(scale).double()
    if shift is not None:
        y = y + torch.from_numpy(shift).double()
    if relu:
        y = torch.relu(y)
    if res is not None:
        y = y + torch.from_numpy(res).double()
    return y.numpy(), raw.numpy()


def bf16_round(a):
    """fp32 -> bfloat16 (round to nearest even) -> fp32: what enters the bf16 matrix cores."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def e2e_case():
    """The committed S=1/F=64/J=16/B=1 case: inputs, regenerated parameters (checked against the
    committed checksum), expected outputs."""
    from densereg_amd.data.synthetic import make_crops
    from oracle import net, pose
    from oracle.graph import NetConfig, trainable_names
    g = golden('e2e_s1f64.npz')
    cfg = NetConfig(1, 64, 16)
    calib = pose.norm_dm(*[make_crops(4, 'icvl', seed=5)[i] for i in (0, 3)])
    params = net.make_test_params(cfg, calib, seed=7)
    names = trainable_names(cfg)
    psum = np.array([np.abs(params[n]).sum(dtype=np.float64) for n in names[:8]] +
                    [sum(np.abs(params[n]).sum(dtype=np.float64) for n in names)])
    np.testing.assert_allclose(psum, g['param_checksum'], rtol=1e-5,
                               err_msg='regenerated test parameters drifted from the committed checksum')
    return cfg, params, g


# ---- raw access to library-owned buffers (flat gradient / parameter views) ----------------------
def _flat_rw(be, addr, n):
    """numpy-like accessor pair (read(), write(a)) over n float32 at device address `addr`."""
    if be.name == 'emu':
        arr = np.ctypeslib.as_array((C.c_float * n).from_address(addr))
        return (lambda: arr.copy()), (lambda a: arr.__setitem__(slice(None), a))
    from densereg_amd.engine import _as_tensor
    t = _as_tensor(addr, n, be.device)

    def write(a):
        t.copy_(be.torch.from_numpy(np.ascontiguousarray(a, np.float32)))
        be.sync()
    return (lambda: (be.sync(), t.cpu().numpy())[1]), write


def flat_grads_by_name(be, h, cfg):
    from oracle.graph import param_specs
    addr, n = h.flat('grad')
    flat = _flat_rw(be, addr, n)[0]()
    out, off = {}, 0
    for name, shape, tr in param_specs(cfg):
        if tr:
            cnt = int(np.prod(shape))
            out[name] = flat[off:off + cnt].reshape(shape).copy()
            off += cnt
    assert off == n
    return out


def grad_metrics(g, ref):
    """Per-tensor agreement of two gradient sets {name: array}: max-norm error (of the tensor's largest element), relative L2 error
    and cosine -- the max-norm sees a single ReLU / max-pool switch, the L2 and cosine see a systematic few-per-cent error of a whole
    tensor that the max-norm bar would let through.  Returns (names, max_err, rel_l2, cosine) as arrays (fp64 accumulation)."""
    names = [n for n in ref if n in g]
    mx, l2, cs = [], [], []
    for n in names:
        a, b = np.asarray(g[n], np.float64).ravel(), np.asarray(ref[n], np.float64).ravel()
        nb = np.linalg.norm(b)
        mx.append(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))
        l2.append(np.linalg.norm(a - b) / (nb + 1e-300))
        cs.append(float(a @ b) / (np.linalg.norm(a) * nb + 1e-300))
    return names, np.array(mx), np.array(l2), np.array(cs)
