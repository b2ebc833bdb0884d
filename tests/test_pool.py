"""Max-pool forward with recorded arg-max + gather backward (kernels_misc.h maxpool_kernel, train_kernels.h maxpool_bwd_kernel)
against torch's max_pool2d / autograd in fp64 on -inf padded inputs laid out by the TF 'SAME' rule (ops.max_pool,
network/slim/ops.py:640-669: extra padding on the bottom / right, padding never wins the max).  Inputs are full of exact ties
(ReLU zeros, values on a coarse grid): the gradient of a window goes to its FIRST maximum in scan order, on both sides."""
import numpy as np
import pytest

BACKENDS = [pytest.param('emu'), pytest.param('gpu', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def be(request):
    return request.getfixturevalue(request.param)


def _ref(x, dy, k):
    import torch
    import torch.nn.functional as F
    B, H, W, C = x.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    th, tw = max((Ho - 1) * 2 + k - H, 0), max((Wo - 1) * 2 + k - W, 0)
    xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2).requires_grad_(True)
    xp = F.pad(xt, (tw // 2, tw - tw // 2, th // 2, th - th // 2), value=float('-inf'))
    y = F.max_pool2d(xp, k, 2)
    (y * torch.from_numpy(dy.astype(np.float64)).permute(0, 3, 1, 2)).sum().backward()
    return y.detach().permute(0, 2, 3, 1).numpy(), xt.grad.permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize('k', [2, 3])
def test_maxpool_forward_and_gather_backward(be, k):
    shapes = [(2, 8, 8, 8), (1, 9, 15, 12), (3, 2, 2, 4), (1, 1, 5, 4)]
    if be.name == 'gpu':
        shapes += [(40, 64, 64, 64), (40, 32, 32, 128)]
    for i, (B, H, W, C) in enumerate(shapes):
        rng = np.random.default_rng(100 * k + i)
        x = np.maximum(np.round(rng.standard_normal((B, H, W, C)) * 2) / 2, 0).astype(np.float32)     # ~half zeros, the rest on a 0.5 grid
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        dy = rng.standard_normal((B, Ho, Wo, C)).astype(np.float32)
        y_ref, dx_ref = _ref(x, dy, k)
        for acc in (0, 1):
            seed = rng.standard_normal(x.shape).astype(np.float32)
            d_x, d_dy = be.dev(x), be.dev(dy)
            d_y, d_dx = be.dev(np.full((B, Ho, Wo, C), -777.0, np.float32)), be.dev(seed)
            rc = be.dbg.dr_dbg_maxpool(B, H, W, C, k, be.ptr(d_x), be.ptr(d_y), be.ptr(d_dy), be.ptr(d_dx), acc, be.stream)
            assert rc == 0, rc
            be.sync()
            np.testing.assert_array_equal(be.host(d_y).reshape(y_ref.shape), y_ref.astype(np.float32))
            want = dx_ref + (seed.astype(np.float64) if acc else 0.0)
            # a pixel collects at most four windows: sums of <= 5 fp32 terms
            np.testing.assert_allclose(be.host(d_dx).reshape(x.shape), want, rtol=0, atol=4e-6 * max(1.0, np.abs(want).max()))


def test_maxpool_rejects_bad_arguments(be):
    d = be.dev(np.zeros(64, np.float32))
    assert be.dbg.dr_dbg_maxpool(1, 4, 4, 6, 2, be.ptr(d), be.ptr(d), be.ptr(d), be.ptr(d), 0, be.stream) != 0      # C % 4
    assert be.dbg.dr_dbg_maxpool(1, 4, 4, 4, 5, be.ptr(d), be.ptr(d), be.ptr(d), be.ptr(d), 0, be.stream) != 0      # k
    assert be.dbg.dr_dbg_maxpool(1, 4, 4, 4, 2, None, be.ptr(d), be.ptr(d), be.ptr(d), 0, be.stream) != 0
