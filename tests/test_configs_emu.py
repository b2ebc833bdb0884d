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
