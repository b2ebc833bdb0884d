"""Process-wide flags with the reference's names and defaults (``model/hourglass_um_crop_tiny.py:29-60``).

The reference defines them with ``tf.app.flags`` and reads them as globals at class-definition and
graph-build time.  README and BASELINE use ``--fea_num`` where the code defines ``--num_fea``
(``readme.md:19,36-38`` vs ``:57``): both spellings are accepted.
"""
from __future__ import annotations

import argparse


def _bool(v):
    if isinstance(v, bool):
        return v
    if str(v).lower() in ('1', 'true', 't', 'yes', 'y'):
        return True
    if str(v).lower() in ('0', 'false', 'f', 'no', 'n'):
        return False
    raise argparse.ArgumentTypeError('boolean expected, got %r' % v)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description='dense-regression hand-pose engine (MI355X)')
    p.add_argument('--num_gpus', type=int, default=1, help='how many gpu to be used')
    p.add_argument('--batch_size', type=int, default=40, help='batch size')
    p.add_argument('--debug_level', type=int, default=1, help='the higher, the more saved to summary')
    p.add_argument('--sub_batch', type=int, default=5, help='micro-batches accumulated per optimizer step')
    p.add_argument('--pid', type=int, default=0, help='for msra person id')
    p.add_argument('--is_train', type=_bool, default=True, help='True for training, False for testing')
    p.add_argument('--net_module', default='um_v1', help='the module containing the network architecture')
    p.add_argument('--is_aug', type=_bool, default=True, help='whether to augment data')
    p.add_argument('--dataset', default='nyu', choices=['nyu', 'icvl', 'msra'], help='the dataset to conduct experiments')
    p.add_argument('--epoch', type=int, default=80, help='number of epoches')
    p.add_argument('--num_stack', type=int, default=2, help='number of stacked hourglass')
    p.add_argument('--num_fea', '--fea_num', dest='num_fea', type=int, default=128, help='number of feature maps in hourglass')
    p.add_argument('--kernel_size', type=int, default=3, help='kernel size for the residual module')
    # additions of this implementation (no dataset / checkpoint ships with the repo)
    p.add_argument('--in_hw', type=int, default=128, choices=[128, 256, 512],
                   help='side of the square input crop; the network accepts 128 / 256 / 512 (um_v1.py:99-104), the reference model '
                   'class hard-codes 128 (hourglass_um_crop_tiny.py:82-87); maps are in_hw/4')
    p.add_argument('--max_steps', type=int, default=0, help='stop after this many optimizer steps (0 = the reference schedule)')
    p.add_argument('--num_frames', type=int, default=0, help='test: number of synthetic frames (0 = dataset exact_num)')
    p.add_argument('--seed', type=int, default=20240)
    p.add_argument('--synthetic_crops', type=int, default=4000,
                   help='size of the synthetic stand-in dataset (no --data_dir): batch index i draws the seeded crops of index i modulo '
                   'synthetic_crops // batch_size, generated once and kept on the host -- epochs over a finite set, as with the record files')
    p.add_argument('--groups', type=int, default=-1,
                   help='training: run the sub_batch micro-steps of an accumulation window as one pass of launches (dr_set_groups): '
                   '-1 = where it pays (parallel.window_groups), 0/1 = one micro-step per pass, sub_batch = always')
    p.add_argument('--precision', default='f32', choices=['f32', 'bf16'],
                   help='matrix-core arithmetic of the convolutions (dr_set_precision); the reference is fp32')
    p.add_argument('--data_dir', default='', help='dataset root holding the TFRecord shards (exp/data/<dataset>/ in the reference); '
                   'empty = seeded synthetic crops')
    p.add_argument('--restore_step', type=int, default=0, help='restore <train_dir>/model.ckpt-<step> (TF V2 checkpoint) before training / testing; 0 = none when training, '
                   'model.ckpt--1 if present when testing (the reference tests with step -1)')
    p.add_argument('--save_every', type=int, default=0, help='train: write <train_dir>/model.ckpt-<step> every N optimizer steps and at the end (0 = never)')
    return p


FLAGS = build_parser().parse_args([])


def parse(argv=None):
    global FLAGS
    ns = build_parser().parse_args(argv)
    FLAGS.__dict__.update(ns.__dict__)
    return FLAGS
