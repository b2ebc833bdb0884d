"""Generate the committed golden vectors from the CPU oracle (run here, in the dev container).

The reference (Py2.7 + TF1.3) can be neither imported nor built in this image and ships no test
vectors, so these fixtures come from ``oracle/`` (PARITY UNPINNED, see oracle/__init__.py); they pin
the oracle against drift and give the GPU tests reference outputs that do not depend on recomputing
the oracle on the GPU box.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from densereg_amd.data.synthetic import make_crops  # noqa: E402
from oracle import net, pose, train  # noqa: E402
from oracle.graph import NetConfig, trainable_names  # noqa: E402


def vote_cases():
    """Five crafted (sample) cases for the vote, J=4: planted peaks / exact ties / negative weights /
    out-of-range re-projection / all-background crop."""
    rng = np.random.default_rng(11)
    B, J, m = 5, 4, 32
    dm, _, cfgs, coms, _ = make_crops(B, 'icvl', seed=77)
    ndm = pose.norm_dm(dm, coms)
    tiny = ndm[:, ::4, ::4, :]
    hm = rng.uniform(0.0, 0.2, (B, m, m, J)).astype(np.float32)
    hm3 = rng.uniform(0.0, 0.2, (B, m, m, J)).astype(np.float32)
    um = rng.uniform(-0.3, 0.3, (B, m, m, 3 * J)).astype(np.float32)
    # 0: planted peaks on foreground pixels
    fg = np.argwhere(tiny[0, :, :, 0] >= -0.99)
    for j in range(J):
        for k in range(6):
            y, x = fg[rng.integers(len(fg))]
            hm3[0, y, x, j] = 0.9 - 0.05 * k
            hm[0, y, x, j] = 0.8
    # 1: exact ties everywhere (constant maps)
    hm[1], hm3[1] = 0.5, 0.25
    # 2: negative candidate weights (hm < 0 but refined map still positive)
    hm[2] = rng.uniform(-0.9, -0.1, (m, m, J)).astype(np.float32)
    hm3[2] = rng.uniform(0.1, 1.0, (m, m, J)).astype(np.float32)
    # 3: offsets that re-project outside the map -> weight 0 -> zero kernel mass guard
    um[3] = 50.0
    # 4: all-background crop
    ndm[4] = -1.0
    # stored as float16 in the fixture: round first so the expected outputs belong to the stored inputs
    q = lambda a: a.astype(np.float16).astype(np.float32)
    return dict(hm=q(hm), hm3=q(hm3), um=q(um), dm_norm=ndm, cfg=cfgs, com=coms)


def main():
    rng = np.random.default_rng(3)
    # ---- (1) per-kernel conv cases --------------------------------------------------------------
    import torch
    import torch.nn.functional as F
    conv = {}
    for i, (B, H, W, Cin, Cout, k) in enumerate([(2, 6, 5, 65, 65, 3), (1, 8, 8, 131, 128, 1), (3, 4, 4, 20, 70, 3)]):
        x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
        w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
        scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        shift = rng.standard_normal(Cout).astype(np.float32)
        res = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
        y = F.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(w).double().permute(3, 2, 0, 1),
                     padding=k // 2).permute(0, 2, 3, 1)
        y = torch.relu(y * torch.from_numpy(scale).double() + torch.from_numpy(shift).double()) + torch.from_numpy(res).double()
        conv.update({'x%d' % i: x, 'w%d' % i: w, 'scale%d' % i: scale, 'shift%d' % i: shift, 'res%d' % i: res,
                     'y%d' % i: y.numpy().astype(np.float32)})
    np.savez_compressed(os.path.join(HERE, 'conv_cases.npz'), **conv)

    # ---- (2) end-to-end S=1 F=64 J=16 B=1 (config-1 stand-in) -----------------------------------
    cfg = NetConfig(1, 64, 16)
    dm, poses, cfgs, coms, _ = make_crops(1, 'icvl', seed=20240)
    ndm = pose.norm_dm(dm, coms)
    calib = pose.norm_dm(*[make_crops(4, 'icvl', seed=5)[i] for i in (0, 3)])
    params = net.make_test_params(cfg, calib, seed=7)
    ep = net.forward_eval(cfg, params, ndm)
    hm, hm3, um = ep['hm_outs'][-1], ep['hm3_outs'][-1], ep['um_outs'][-1]
    xyz = pose.estimate_pose_mm(hm, hm3, um, ndm, cfgs, coms)
    names = trainable_names(cfg)
    psum = np.array([np.abs(params[n]).sum(dtype=np.float64) for n in names[:8]] +
                    [sum(np.abs(params[n]).sum(dtype=np.float64) for n in names)])
    losses, grads, _, _ = train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms)
    gsel = [names[0], names[1], names[2], names[len(names) // 2], names[-2], names[-1]]
    np.savez_compressed(os.path.join(HERE, 'e2e_s1f64.npz'), dm=dm,
                        pose=poses, cfg=cfgs, com=coms, param_checksum=psum,
                        hm=hm[:, ::2, ::2], hm3=hm3[:, ::2, ::2], um=um[:, ::2, ::2], xyz=xyz,
                        losses=np.array([losses['hm'], losses['hm3'], losses['um'], losses['reg']], np.float64),
                        grad_names=np.array(gsel), grad_abs_sum=np.array([np.abs(grads[n]).sum(dtype=np.float64) for n in gsel]),
                        grad_sum=np.array([grads[n].sum(dtype=np.float64) for n in gsel]))

    # ---- (3) crafted vote cases -----------------------------------------------------------------
    vc = vote_cases()
    om = pose.resume_om(vc['hm3'], vc['um'])
    tiny = vc['dm_norm'][:, ::4, ::4, :]
    n, dbg = pose.xyz_estimation(vc['hm'], om, vc['hm3'], tiny, vc['cfg'], vc['com'], return_debug=True)
    xyz = pose.unnorm_xyz_pose(n.reshape(5, -1), vc['com'])
    np.savez_compressed(os.path.join(HERE, 'vote_cases.npz'), hm=vc['hm'].astype(np.float16), hm3=vc['hm3'].astype(np.float16),
                        um=vc['um'].astype(np.float16), tiny=tiny, cfg=vc['cfg'], com=vc['com'], xyz=xyz, idx=dbg['idx'],
                        w=dbg['w'])
    for f in ('conv_cases.npz', 'e2e_s1f64.npz', 'vote_cases.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'kB')


if __name__ == '__main__':
    main()
