"""CPU-side (emulator) coverage of the joint counts / input sizes of BASELINE configs 4 and 5 on narrow networks: the
graph builder, buffer plan, ragged channel slices and the vote's joint chunking for J=21 (MSRA, ``data/msra.py:13-17``),
through the same C ABI the GPU tests use.  The full-width forms run on the GPU (tests/test_gpu_configs.py)."""
import numpy as np

from tests.test_gpu_configs import _case
from tests.test_train_parity import _run_step


def test_msra_j21_forward_vote_and_train_step_on_the_emulator(emu):
    from oracle import net, pose
    cfg, params, ndm, poses, cfgs, coms = _case(1, 16, 21, 1, 'msra')
    h = emu.handle(cfg, 1)
    h.load_params(params)
    h.call('dr_finalize_params', None)
    hm, hm3, um = emu.forward_eval(h, ndm)
    assert hm.shape == (1, 32, 32, 21) and um.shape == (1, 32, 32, 63)
    ep = net.forward_eval(cfg, params, ndm)
    for got, key in ((hm, 'hm_outs'), (hm3, 'hm3_outs'), (um, 'um_outs')):
        assert np.abs(got - ep[key][-1]).max() < 5e-4, key
    xyz = emu.infer(h, ndm, cfgs, coms)
    ref = pose.estimate_pose_mm(ep['hm_outs'][-1], ep['hm3_outs'][-1], ep['um_outs'][-1], ndm, cfgs, coms)
    assert xyz.shape == (1, 63) and pose.mean_jnt_error(xyz, ref) <= 0.1
    h.close()
    # with an injected dropout mask: the um_full convs' backward (mask x2 and bias column sums) comes out of their readers'
    # dgrad epilogues (conv_igemm.h, bst_act)
    rng = np.random.default_rng(0)
    masks = [rng.integers(0, 2, (1, 32, 32, 512)).astype(np.uint8) for _ in range(2)]
    h, _ = _run_step(emu, cfg, params, ndm, poses, cfgs, coms, masks)
    h.close()


def test_train_step_with_the_x3_kernels_forced_on_every_layer_on_the_emulator(emu):
    """The product's DEFAULT kernel selection runs the x3 family on the big layers (conv_x3.h, conv_x3h.h, conv_wgrad_x3.h); the emulator
    suite otherwise forces the fp32-MFMA kernels for speed (tests/common.py).  This test keeps the handle-level wiring of the x3 family
    covered without a GPU: one training micro-step of a small network (S=1, F=32, J=4) with dr_dbg_force_x3(2) -- the launcher rule, the
    [chunk][tap][Np][3][16] weight planes of pack_all_kernel, the 128-row tiles' statistics rows, the halo kernel on the 32x32 3x3 layers,
    the hidden weight copies, the x3 weight gradients -- against the oracle like every other training step."""
    from tests.test_train_parity import _case as _train_case
    cfg, params, ndm, poses, cfgs, coms = _train_case(1, 32, 4, 1)
    n_h = emu.dbg.dr_dbg_x3h_launches()
    try:
        assert emu.dbg.dr_dbg_force_x3(2) == 0
        h, _ = _run_step(emu, cfg, params, ndm, poses, cfgs, coms, None)
        h.close()
    finally:
        emu.dbg.dr_dbg_force_x3(emu.x3_default)
    assert emu.dbg.dr_dbg_x3h_launches() > n_h, 'the halo kernel did not run on the 3x3 layers of the 32x32 maps'
