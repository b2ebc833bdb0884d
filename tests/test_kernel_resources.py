"""Static guard on the product build (no GPU needed): hipcc's resource remarks for gfx950 must show NO kernel spilling registers
to scratch, and the hot conv kernels must keep the occupancy their tile choice assumes (five waves per SIMD for the 64x128 /
64x64 / 128x32 tiles: M = 40960 rows fall on the chip in whole rounds of five workgroups per CU, conv_igemm.h).  A change that
adds a few live registers to a shared epilogue shows up here before it shows up as a slower step (it did: two run-time branches in
conv_epilogue.inc once cost the bf16 kernels 20-216 bytes of scratch per lane and the fp32 step 1 %)."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_no_kernel_spills_and_hot_kernels_keep_their_occupancy():
    import kernel_resources
    rows = kernel_resources.collect()
    assert len(rows) > 60
    # ONE exception, bounded: the bf16 64x128 kernel that stages an fp32-stored A operand, in the instantiation whose epilogue
    # comes in three copies (XB = 2: bf16-stored raw outputs / gradients of the training step, conv_epilogue.inc), at its 96
    # registers keeps five values of the prologue (lane / row coordinates) in 24 bytes of scratch across the K loop -- stored
    # before the loop, reloaded behind it, nothing inside it (ISA read, profiles/r04_experiments.md section 9).  Anything more, or
    # any other kernel -- the one-copy instantiations the eval path runs included -- fails.
    allowed = {'dr::conv_igemm_kernel<64, 128, 2, 2, 0, 16, 0, 1, 1, 2, 32>': 32}
    spills = [(n, r['scratch']) for n, r in rows if r.get('scratch', 0) > allowed.get(n, 0)]
    assert not spills, spills
    occ = {n: r['occ'] for n, r in rows}
    for name in ('dr::conv_igemm_kernel<64, 128, 2, 2, 0, 16, 1, 0, 1, 0, 32>', 'dr::conv_igemm_kernel<64, 64, 2, 2, 0, 16, 1, 0, 1, 0, 32>',
                 'dr::conv_igemm_kernel<128, 32, 4, 1, 0, 16, 0, 0, 1, 0, 32>', 'dr::conv_igemm_kernel<64, 128, 2, 2, 0, 16, 0, 1, 1, 0, 32>',
                 'dr::conv_igemm_kernel<64, 144, 4, 1, 0, 16, 0, 0, 1, 0, 16>', 'dr::conv_igemm_kernel<64, 160, 4, 1, 0, 16, 0, 0, 1, 0, 16>'):
        assert occ.get(name) == 5, (name, occ.get(name))
    assert occ.get('dr::conv_igemm_kernel<64, 80, 4, 1, 0, 16, 0, 0, 1, 0, 16>', 0) >= 5
    assert occ['dr::conv_wgrad_kernel<128>'] >= 3 and occ['dr::bn_train_apply_kernel<0, 0>'] >= 5
    # the x3 kernels (round 5): eight waves per workgroup need four waves per SIMD = at most 128 registers; the four-wave forms two
    # (template arguments: BM, BN, LO, RING, NW, WM, BD, PF, ABL -- BD = 1: weight tiles by the hidden LDS-DMA, the product's 128-column
    # kernel; PF / ABL: the measurement variants of round 6, 0 in the product)
    for name in ('dr::conv_x3_kernel<128, 128, 1, 0, 8, 2, 1, 0, 0>', 'dr::conv_x3_kernel<128, 128, 1, 0, 8, 2, 0, 0, 0>'):
        assert occ.get(name) == 4, (name, occ.get(name))
    assert occ.get('dr::conv_x3_kernel<128, 96, 1, 0, 4, 4, 0, 0, 0>', 0) >= 2 and occ.get('dr::conv_x3_kernel<128, 64, 1, 0, 4, 2, 0, 0, 0>', 0) >= 3
    # the 160-column tile (round 6): four waves of 32 x 160 -- ten accumulator tiles per wave, two waves per SIMD (<= 256 registers, no scratch:
    # the spill check above is what caught an epilogue change that cost this kernel 76 bytes per lane)
    assert occ.get('dr::conv_x3_kernel<128, 160, 1, 0, 4, 4, 1, 0, 0>', 0) >= 2
    # the 3x3 halo kernels (round 6, conv_x3h.h: BN, log2 W, NW, WM): the same budgets, and two workgroups per CU by LDS (<= 80 KB each)
    lds = {n: r.get('lds', 0) for n, r in rows}
    for name, want in (('dr::conv_x3h_kernel<128, 5, 8, 2>', 4), ('dr::conv_x3h_kernel<128, 4, 8, 2>', 4), ('dr::conv_x3h_kernel<96, 5, 4, 4>', 2),
                       ('dr::conv_x3h_kernel<64, 5, 4, 2>', 2)):
        assert occ.get(name, 0) >= want and 0 < lds.get(name, 0) <= 80 * 1024, (name, occ.get(name), lds.get(name))
    assert occ.get('dr::conv_wgrad_x3_kernel<128, 4>', occ.get('dr::conv_wgrad_x3_kernel<128>', 0)) >= 2
