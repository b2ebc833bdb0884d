"""Parity of the HIP forward path (conv kernel, whole network, vote) against the CPU oracle and the
committed golden vectors, through the C ABI.

Every test runs twice: ``[emu]`` = the same kernel sources on the host-fiber emulator (CPU, catches
index/layout bugs without a device) and ``[gpu]`` = the real library on an MI355X (``-m gpu``).
Tolerances: fp32 everywhere; conv vs an fp64 reference 2e-5 relative to the tensor scale (fp32
accumulation-order noise), maps 2e-4 absolute, xyz BIT-EXACT on identical maps (the vote's exp is one fixed fp32 sequence on both sides),
and BASELINE.json's <= 0.1 mm mean-joint-error delta end to end.
"""
import numpy as np
import pytest

from tests.common import bf16_round, e2e_case, golden, ref_conv2d

BACKENDS = [pytest.param('emu'), pytest.param('gpu', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def be(request):
    return request.getfixturevalue(request.param)


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


# ---------------------------------------------------------------------------------------------
# conv kernel
# ---------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, H, W, Cin, Cout, k      (ragged channel counts of the real net: 65, 131, 515, 80, 5J ...)
    (1, 8, 8, 16, 32, 1),
    (2, 6, 5, 65, 65, 3),
    (1, 16, 16, 131, 128, 1),
    (3, 4, 4, 20, 70, 3),
    (2, 2, 2, 128, 64, 1),
    (1, 8, 8, 160, 256, 1),
    (1, 4, 4, 515, 512, 1),
    (1, 9, 7, 80, 48, 3),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_fused_epilogue(be, case):
    B, H, W, Cin, Cout, k = case
    rng = np.random.default_rng(hash(case) % 2**31)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    mask = rng.uniform(-1, 1, B * H * W).astype(np.float32) if k == 1 else None
    y, st = be.conv2d(x, w, scale, shift, True, res, mask, -0.5, want_stats=True)
    yr, raw = ref_conv2d(x, w, scale, shift, True, res, mask, -0.5)
    assert _rel(y, yr) < 2e-5
    np.testing.assert_allclose(st[0], raw.sum((0, 1, 2)), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(st[1], (raw ** 2).sum((0, 1, 2)), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('tile', [6, 7, 8], ids=['splitk32x32', 'narrow64x96', 'narrow64x160'])
@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_splitk_kernel_fused_epilogue(be, case, tile):
    """The kernels whose waves split K and reduce through LDS, with every fused epilogue feature incl. the BatchReNorm
    statistics: conv_splitk_kernel (tile 6: 32x32 output tile, four-way split; partial rows come four to a workgroup; cases
    with fewer K-tiles than waves -- 1x1, Cin = 16 -- are part of the list) and the narrow-output tiles of conv_igemm.h (7 /
    8: 64 rows x 96 / 160 columns, waves = 2 row halves x 2 K halves, partner sums parked in LDS; forced on every case of the
    list, so also on outputs wider than one column block and narrower than a quarter of it)."""
    B, H, W, Cin, Cout, k = case
    rng = np.random.default_rng(hash(case) % 2**31 + 1)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    mask = rng.uniform(-1, 1, B * H * W).astype(np.float32) if k == 1 else None
    try:
        assert be.dbg.dr_dbg_force_tile(tile) == 0
        y, st = be.conv2d(x, w, scale, shift, True, res, mask, -0.5, want_stats=True)
    finally:
        be.dbg.dr_dbg_force_tile(-1)
    yr, raw = ref_conv2d(x, w, scale, shift, True, res, mask, -0.5)
    assert _rel(y, yr) < 2e-5
    np.testing.assert_allclose(st[0], raw.sum((0, 1, 2)), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(st[1], (raw ** 2).sum((0, 1, 2)), rtol=1e-4, atol=1e-4)


X3_CASES = CONV_CASES + [
    # (B, H, W, Cin, Cout, k): more than one 128-row / 128-column block, ragged rows, K tails of 1..3 channels, Cout beyond a block
    (2, 12, 11, 64, 128, 3), (1, 16, 16, 256, 256, 1), (3, 7, 7, 33, 130, 3), (1, 20, 13, 515, 200, 1), (2, 8, 8, 18, 40, 3),
    # whole 128-column blocks: what conv_p3.h takes (mode 6) -- ragged Cin (chunks of 16 with zero pads), masked rows, ragged last row
    # block, two column blocks, two K-tiles only (the shortest ring), a K-tile count of every residue mod 3
    (1, 9, 10, 40, 128, 3), (2, 7, 9, 131, 256, 1), (1, 16, 8, 32, 128, 1), (1, 11, 12, 78, 128, 3), (1, 13, 10, 64, 128, 1),
    # 3x3 on 16- / 32-pixel-wide images whose 128-row tiles are whole image rows: what conv_x3h.h takes (the haloed tile resident in LDS) --
    # one tile per image (every halo row outside), several tiles per image (halo rows from the neighbours), ragged Cin, one chunk only
    (1, 8, 16, 40, 128, 3), (2, 4, 32, 24, 128, 3), (2, 8, 32, 35, 256, 3), (1, 16, 16, 64, 128, 3), (3, 12, 32, 16, 128, 3),
    # ... on its 96-column (65..96 channels), 64-column and ragged 128-column tiles
    (1, 4, 32, 78, 78, 3), (2, 8, 16, 65, 65, 3), (1, 8, 32, 64, 64, 3), (1, 4, 32, 20, 200, 3),
]


@pytest.mark.parametrize('case', X3_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_x3_fp32_accurate_products_on_the_bf16_matrix_cores(be, case):
    """conv_x3.h: every operand split into three bf16 terms, six cross products per fp32 product on v_mfma_f32_32x32x16_bf16 -- the
    SAME 2e-5 bar against the fp64 definition as the fp32-MFMA kernels, every fused epilogue feature, and an error no larger than
    a small multiple of the fp32 kernel's own on the same problem (so the layer's numbers are fp32 numbers whichever kernel ran)."""
    B, H, W, Cin, Cout, k = case
    rng = np.random.default_rng(hash(case) % 2**31 + 9)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    x *= np.exp(rng.uniform(-6, 6, x.shape)).astype(np.float32)             # operands across twelve binades: every split term matters
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    mask = rng.uniform(-1, 1, B * H * W).astype(np.float32) if k == 1 else None
    yr, raw = ref_conv2d(x, w, scale, shift, True, res, mask, -0.5)
    outs = {}
    halo = k == 3 and W in (16, 32) and (H * W) % 128 == 0      # conv_x3h.h takes the layer in mode 2
    for mode in (0, 2, 4, 5, 6, 7):                          # fp32 matrix cores | x3 (default: eight waves, halo kernel where it applies) | three-stage LDS ring | four waves | P3-stored input | no halo kernel
        try:
            assert be.dbg.dr_dbg_force_x3(mode) == 0
            n_p3, n_h = be.dbg.dr_dbg_p3_launches(), be.dbg.dr_dbg_x3h_launches()
            outs[mode] = be.conv2d(x, w, scale, shift, True, res, mask, -0.5, want_stats=True)
            # (whole 128-column blocks and at least two K-tiles: conv_p3_kernel itself must have run in mode 6, and only there; the halo
            # kernel in mode 2, and in mode 6 where conv_p3_kernel does not apply)
            p3 = mode == 6 and (-(-Cout // 32) * 32) % 128 == 0 and k * k * -(-Cin // 16) >= 2
            assert be.dbg.dr_dbg_p3_launches() - n_p3 == (1 if p3 else 0)
            assert be.dbg.dr_dbg_x3h_launches() - n_h == (1 if halo and (mode == 2 or (mode == 6 and not p3)) else 0)
        finally:
            be.dbg.dr_dbg_force_x3(be.x3_default)
    e32, e3 = _rel(outs[0][0], yr), _rel(outs[2][0], yr)
    rms = lambda y: float(np.sqrt(np.mean((y - yr) ** 2)) / (np.abs(yr).max() + 1e-12))
    r32, r3 = rms(outs[0][0]), rms(outs[2][0])
    print('conv_x3 vs fp64: max %.2e rms %.2e | fp32-MFMA kernel: max %.2e rms %.2e' % (e3, r3, e32, r32))
    assert e3 < 2e-5, (e3, e32)
    # measured on MI355X (profiles/r05_x3_microbench.md): rms equal to the fp32 kernel's (4.2e-8 vs 4.0e-8 on 3x3 256->256, 1.9e-8 vs
    # 5.5e-8 on 1x1 512->512), the maximum within 3x; the emulator's strictly sequential fma chain makes the fp32 kernel look better
    assert r3 < 3 * r32 + 2e-8 and e3 < 6 * e32 + 2e-7, (e3, e32, r3, r32)
    # every variant multiplies the same planes in the same K order: the same bits, whatever the staging and the wave layout
    np.testing.assert_array_equal(outs[2][0], outs[4][0])
    np.testing.assert_array_equal(outs[2][0], outs[5][0])
    # conv_x3h.h (mode 2 where it applies) against conv_x3_kernel on the same layer (mode 7): the same planes, the same product order
    np.testing.assert_array_equal(outs[2][0], outs[7][0])
    np.testing.assert_array_equal(outs[2][1], outs[7][1])
    # conv_p3.h: the input split ONCE by p3_split_kernel, both tiles by LDS-DMA -- the same planes, the same product order
    np.testing.assert_array_equal(outs[2][0], outs[6][0])
    np.testing.assert_array_equal(outs[2][1], outs[6][1])
    for y, st in outs.values():
        np.testing.assert_allclose(st[0], raw.sum((0, 1, 2)), rtol=1e-4, atol=1e-4 * float(np.abs(raw).max()) * raw[..., 0].size ** 0.5)
        np.testing.assert_allclose(st[1], (raw ** 2).sum((0, 1, 2)), rtol=1e-4)


MF16_CASES = [            # (B, H, W, Cin, Cout, k): output widths of the 16-column tiles, ragged and full, odd images, short / long K
    (1, 9, 7, 156, 78, 1), (2, 5, 6, 78, 78, 3), (1, 8, 8, 131, 65, 1), (1, 7, 9, 65, 65, 3), (2, 6, 6, 64, 80, 1),
    (1, 5, 5, 170, 131, 1), (1, 6, 7, 33, 142, 3), (2, 4, 9, 259, 129, 1), (1, 8, 5, 78, 156, 1), (1, 3, 3, 19, 160, 3), (3, 11, 3, 4, 145, 1),
]


@pytest.mark.parametrize('case', MF16_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_16_column_mfma_tiles(be, case):
    """The fp32 tiles on v_mfma_f32_16x16x4_f32 (conv_igemm.h, MF = 16: 64 rows x 80 / 144 / 160 columns, what the launcher
    now selects for 65..80 and 129..160 output channels on grids that fill the chip) with every fused epilogue feature -- scale /
    shift, ReLU, residual, input row mask, BatchReNorm statistics -- against the fp64 definition, next to the heuristic's choice
    and the 128x32 tile (DR_CONV_MF16=0's choice) on the same problem."""
    B, H, W, Cin, Cout, k = case
    rng = np.random.default_rng(hash(case) % 2**31 + 5)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    mask = rng.uniform(-1, 1, B * H * W).astype(np.float32) if k == 1 else None
    tile = {80: 9, 144: 10, 160: 11}[-(-Cout // 16) * 16]
    outs = {}
    for t in (-1, tile, 4):
        try:
            assert be.dbg.dr_dbg_force_tile(t) == 0
            outs[t] = be.conv2d(x, w, scale, shift, True, res, mask, -0.5, want_stats=True)
        finally:
            be.dbg.dr_dbg_force_tile(-1)
    yr, raw = ref_conv2d(x, w, scale, shift, True, res, mask, -0.5)
    for t, (y, st) in outs.items():
        assert _rel(y, yr) < 2e-5, t
        np.testing.assert_allclose(st[0], raw.sum((0, 1, 2)), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(st[1], (raw ** 2).sum((0, 1, 2)), rtol=1e-4, atol=1e-4)
    # (at these sizes the heuristic (-1) still prefers the split-K kernel: grids of a few workgroups; at B=40 x 32x32 it selects
    # the 16-column tiles -- the whole-network B=40 parity tests of test_gpu_configs.py / test_gpu_fullsize.py run through them)


def test_conv_transpose_detecting(be):
    """A = identity-like input with an ASYMMETRIC weight matrix: catches a swapped MFMA C/D layout."""
    Cin = Cout = 64
    x = np.zeros((1, 8, 8, Cin), np.float32)
    for m in range(64):
        x.reshape(64, Cin)[m, m] = 1.0                        # pixel m has a one in channel m
    w = (np.arange(Cin)[:, None] * 100 + np.arange(Cout)[None, :]).astype(np.float32).reshape(1, 1, Cin, Cout)
    y = be.conv2d(x, w)
    np.testing.assert_array_equal(y.reshape(64, Cout), w.reshape(Cin, Cout))     # exact in fp32


def test_conv_golden_vectors(be):
    g = golden('conv_cases.npz')
    for i in range(3):
        y = be.conv2d(g['x%d' % i], g['w%d' % i], g['scale%d' % i], g['shift%d' % i], True, g['res%d' % i])
        assert _rel(y, g['y%d' % i]) < 2e-5


@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
def test_conv_every_tile_shape(be, tile):
    """Each tile configuration of the implicit-GEMM kernel (the heuristic only exercises some per shape)."""
    rng = np.random.default_rng(tile)
    np_needed = {0: 128, 1: 128, 2: 64, 3: 64, 4: 32, 5: 64, 6: 32, 7: 96, 8: 160, 9: 80, 10: 144, 11: 160}[tile]
    # 5 = 64x64 tile with the fat (BK = 64) K-tile, 6 = 32x32 tile, K split over the four waves, 7 / 8 = 64x96 / 64x160, waves =
    # 2 rows x 2 K halves, 9 / 10 / 11 = 64x80 / 64x144 / 64x160 on v_mfma_f32_16x16x4_f32 (four waves = four 16-row groups)
    Cout, Cin, k = np_needed - 3, 37, 3
    x = rng.standard_normal((1, 9, 15, Cin)).astype(np.float32)          # 135 rows: ragged last M tile
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    shift = rng.standard_normal(Cout).astype(np.float32)
    try:
        assert be.dbg.dr_dbg_force_tile(tile) == 0
        y = be.conv2d(x, w, None, shift, True)
    finally:
        be.dbg.dr_dbg_force_tile(-1)
    yr, _ = ref_conv2d(x, w, None, shift, True)
    assert _rel(y, yr) < 2e-5
    # one and two K-tiles (1x1, Cin = 12 and 20): the unrolled-by-two loop with an odd tail; 70 / 130 channels:
    # a short last chunk group of the fat K-tile
    for cin in (12, 20, 70, 130):
        x1 = rng.standard_normal((2, 4, 5, cin)).astype(np.float32)
        w1 = rng.standard_normal((1, 1, cin, Cout)).astype(np.float32)
        try:
            be.dbg.dr_dbg_force_tile(tile)
            y1 = be.conv2d(x1, w1)
        finally:
            be.dbg.dr_dbg_force_tile(-1)
        assert _rel(y1, ref_conv2d(x1, w1)[0]) < 2e-5


@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4])
def test_conv_workgroup_order_with_several_column_blocks(be, tile):
    """The workgroup -> tile mapping deals the N blocks of a row block out back to back (conv_igemm.h, nfast), per XCD when
    the number of row blocks is a multiple of 8 and linearly otherwise: every (row block, column block) pair must be computed
    exactly once -- three column blocks, with 16 / 8 row blocks (XCD form) and with 3 / 2 (linear form)."""
    rng = np.random.default_rng(40 + tile)
    bn = {0: 128, 1: 128, 2: 64, 3: 64, 4: 32}[tile]
    Cout, Cin = 3 * bn - 5, 24
    for shape in ((4, 16, 16), (1, 9, 15)):                              # 1024 rows / 135 rows
        x = rng.standard_normal(shape + (Cin,)).astype(np.float32)
        w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
        shift = rng.standard_normal(Cout).astype(np.float32)
        try:
            assert be.dbg.dr_dbg_force_tile(tile) == 0
            y = be.conv2d(x, w, None, shift, True)
        finally:
            be.dbg.dr_dbg_force_tile(-1)
        assert _rel(y, ref_conv2d(x, w, None, shift, True)[0]) < 2e-5


def test_conv_seeded_shape_sweep(be):
    """A seeded sweep over small random problems -- batch, odd image sides, ragged channel counts, kernel size, every tile the
    shape admits, with / without scale-shift, ReLU, residual, input row mask, statistics -- against the fp64 definition.  Index math
    only (ragged last tiles in M, N and K, the XCD / N-fast workgroup mapping, tap masks at the image border, pad channels as
    poison): the kernels' arithmetic is pinned by the cases above."""
    rng = np.random.default_rng(20240927)
    n_cases = 120 if be.name == "emu" else 96
    for case in range(n_cases):
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(1, 12))
        k = int(rng.choice([1, 3]))
        Cin = int(rng.choice([1, 3, 4, 16, 19, 33, 64, 70, 131]))
        Cout = int(rng.choice([1, 5, 14, 32, 42, 65, 78, 96, 128, 131, 160, 170]))
        np_ = -(-Cout // 32) * 32
        c16 = -(-Cout // 16) * 16
        tiles = [-1, 4, 6] + ([2, 3, 5] if np_ % 64 == 0 else []) + ([0, 1] if np_ % 128 == 0 else []) + ([7] if np_ == 96 else []) + ([8] if np_ == 160 else []) + \
            ([{80: 9, 144: 10, 160: 11}[c16]] if (Cout > 64 and c16 in (80, 144, 160)) else [])
        tile = int(rng.choice(tiles))
        x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
        w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
        scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32) if rng.random() < 0.5 else None
        shift = rng.standard_normal(Cout).astype(np.float32) if rng.random() < 0.5 else None
        relu = bool(rng.random() < 0.5)
        res = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if rng.random() < 0.4 else None
        mask = rng.uniform(-1, 1, B * H * W).astype(np.float32) if (k == 1 and rng.random() < 0.3) else None
        try:
            assert be.dbg.dr_dbg_force_tile(tile) == 0
            y, st = be.conv2d(x, w, scale, shift, relu, res, mask, -0.2, want_stats=True)
        finally:
            be.dbg.dr_dbg_force_tile(-1)
        yr, raw = ref_conv2d(x, w, scale, shift, relu, res, mask, -0.2)
        tag = (case, B, H, W, Cin, Cout, k, tile)
        assert _rel(y, yr) < 2e-5, tag
        np.testing.assert_allclose(st[0], raw.sum((0, 1, 2)), rtol=1e-4, atol=1e-4, err_msg=str(tag))
        np.testing.assert_allclose(st[1], (raw ** 2).sum((0, 1, 2)), rtol=1e-4, atol=1e-4, err_msg=str(tag))


def test_conv_lds_dma_refill_variant(be, monkeypatch):
    """Shapes the LDS-DMA refill (the default; DR_CONV_GLDS=0 = register-staged: global_load_lds, swizzle on the source side, zero page for masked
    chunks) is eligible for -- whole 16-byte channel chunks, incl. a short last chunk group (Cin = 20) -- against the
    reference.  The switch is read once per process: run this file with DR_CONV_GLDS=0 to exercise the other variant
    (tools/gpu/r02_switches.sh does on the GPU; the default run covers the LDS-DMA refill on the same shapes)."""
    rng = np.random.default_rng(11)
    outs = []
    for cin, cout, k, hw in ((64, 128, 3, 8), (20, 64, 1, 6), (256, 64, 3, 4)):
        x = rng.standard_normal((2, hw, hw, cin)).astype(np.float32)
        w = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
        y = be.conv2d(x, w)
        assert _rel(y, ref_conv2d(x, w)[0]) < 2e-5
        outs.append(y)


@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4, 6, 7, 8])
def test_conv_bf16_matrix_core_variant(be, tile):
    """BF kernels (v_mfma_f32_32x32x16_bf16, conv_igemm.h): both operands rounded to bf16 as they are staged, fp32
    accumulation and epilogue.  Reference = the fp64 conv of the bf16-rounded operands, so what is left is fp32
    summation order (products of two bf16 are exact in fp32).  Shapes: ragged Cin on both sides of the 4-channel
    half slot (37, 67, 515-like 35), a short last 32-channel K-tile, a single K-tile, row mask, residual, poison in
    the padding channels."""
    rng = np.random.default_rng(100 + tile)
    np_needed = {0: 128, 1: 128, 2: 64, 3: 64, 4: 32, 6: 96, 7: 96, 8: 160}[tile]
    Cout = np_needed - 3
    try:
        assert be.dbg.dr_dbg_force_tile(tile) == 0 and be.dbg.dr_dbg_force_bf16(1) == 0
        for (cin, k, hw, extras) in ((37, 3, (9, 15), True), (67, 1, (4, 5), True), (35, 3, (5, 4), False), (64, 3, (8, 8), True),
                                     (12, 1, (3, 7), False), (96, 1, (6, 6), False), (6, 3, (4, 4), False)):
            x = rng.standard_normal((2,) + hw + (cin,)).astype(np.float32)
            w = (rng.standard_normal((k, k, cin, Cout)) / np.sqrt(k * k * cin)).astype(np.float32)
            scale = shift = res = mask = None
            if extras:
                scale = (0.5 + rng.random(Cout)).astype(np.float32)
                shift = rng.standard_normal(Cout).astype(np.float32)
                res = rng.standard_normal((2,) + hw + (Cout,)).astype(np.float32)
                if k == 1:                                   # the network masks rows of 1x1 convs only (um_v1.py:146-148)
                    mask = np.where(rng.random((2,) + hw) < 0.3, -1.0, 0.5).astype(np.float32)
            y = be.conv2d(x, w, scale, shift, extras, res, mask, -0.5)
            yr, _ = ref_conv2d(bf16_round(x), bf16_round(w), scale, shift, extras, res, mask, -0.5)
            assert _rel(y, yr) < 2e-5, (cin, k)
            # and it IS the bf16 path: the fp32 result differs by the operand rounding (~2^-9 relative)
            y32, _ = ref_conv2d(x, w, scale, shift, extras, res, mask, -0.5)
            assert 1e-4 < _rel(y, y32) < 3e-2, (cin, k, _rel(y, y32))
    finally:
        be.dbg.dr_dbg_force_tile(-1)
        be.dbg.dr_dbg_force_bf16(0)


def test_bf16_operand_rounding_is_nearest_even(be):
    """The staging path rounds activations with v_cvt_pk_bf16_f32 (the emulator: clang's float -> __bf16), the packing
    kernel rounds weights the same way: round to nearest, ties to even.  Exact ties (a 1 in mantissa bit 15, zeros
    below) through an identity 1x1 conv come out as the even neighbour, bit for bit; so do values just above / below a
    tie, negative numbers, and the weight side (a diagonal of tie values against a one-hot input)."""
    C_ = 32
    k = np.arange(C_, dtype=np.uint32)
    bits = (np.uint32(0x3F800000) | (k << np.uint32(16)) | np.uint32(0x8000))           # 1.xxxxxxx | tie bit
    ties = bits.view(np.float32)
    above = (bits + np.uint32(1)).view(np.float32)                                       # must round up
    below = (bits - np.uint32(1)).view(np.float32)                                       # must round down
    x = np.stack([ties, -ties, above, below]).reshape(1, 2, 2, C_).astype(np.float32)
    eye = np.eye(C_, dtype=np.float32).reshape(1, 1, C_, C_)
    expect_even = ((bits + np.uint32(0x7FFF) + ((bits >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    np.testing.assert_array_equal(bf16_round(ties), expect_even)                         # the reference rounding itself
    try:
        assert be.dbg.dr_dbg_force_bf16(1) == 0
        for tile in (1, 3, 6):
            assert be.dbg.dr_dbg_force_tile(tile) == 0
            y = be.conv2d(x, eye)
            np.testing.assert_array_equal(y.reshape(4, C_), bf16_round(x).reshape(4, C_))
            # weights: diag(ties) against ones -> the rounded diagonal
            w = (np.eye(C_, dtype=np.float32) * ties[None, :]).reshape(1, 1, C_, C_)
            yw = be.conv2d(np.ones((1, 1, 1, C_), np.float32), w)
            np.testing.assert_array_equal(yw.reshape(C_), expect_even)
    finally:
        be.dbg.dr_dbg_force_tile(-1)
        be.dbg.dr_dbg_force_bf16(0)


def test_conv_plain_linear(be):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 5, 5, 24)).astype(np.float32)
    w = rng.standard_normal((3, 3, 24, 40)).astype(np.float32)
    y = be.conv2d(x, w)
    yr, _ = ref_conv2d(x, w)
    assert _rel(y, yr) < 2e-5


# ---------------------------------------------------------------------------------------------
# whole network (config-1 stand-in: ICVL S=1 F=64 B=1) against golden + oracle
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def case1():
    return e2e_case()


def _net_setup(be, cfg, params, B):
    h = be.handle(cfg, B)
    from oracle.graph import param_specs
    assert [(n, tuple(s), t) for n, s, t in h.param_infos()] == [(n, tuple(s), t) for n, s, t in param_specs(cfg)]
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    return h


def test_network_forward_and_vote_config1(be, case1):
    from oracle import net, pose
    from oracle.graph import conv_specs
    cfg, params, g = case1
    h = _net_setup(be, cfg, params, 1)
    assert abs(h.conv_flops_per_crop() / 1e9 - 4.262) < 1e-3
    dm, cfgs, coms = g['dm'], g['cfg'], g['com']
    ndm = be.norm_dm(h, dm, coms)
    np.testing.assert_array_equal(ndm, pose.norm_dm(dm, coms))
    hm, hm3, um = be.forward_eval(h, ndm)
    # committed golden (oracle) maps, stored on a ::2 grid
    assert np.abs(hm[:, ::2, ::2] - g['hm']).max() < 2e-4
    assert np.abs(hm3[:, ::2, ::2] - g['hm3']).max() < 2e-4
    assert np.abs(um[:, ::2, ::2] - g['um']).max() < 2e-4
    # (that forward ran the hourglass bottom as ONE launch, hg_fused.h; with the fusion off every layer's output stays in HBM --
    # and the two forwards agree to fp32 summation order)
    from densereg_amd._lib import DenseRegError
    with pytest.raises(DenseRegError):
        be.read_activation(h, 'Conv_10', (1, 8, 8, 32))             # lower1 @8x8, 3x3: lives only in LDS
    h.call('dr_set_fusion', 0)
    hm_u, hm3_u, um_u = be.forward_eval(h, ndm)
    for a_f, a_u in ((hm, hm_u), (hm3, hm3_u), (um, um_u)):
        assert np.abs(a_f - a_u).max() <= 2e-5 * max(1.0, float(np.abs(a_u).max()))
    # every conv output against the oracle's record
    rec = {}
    net.forward_eval(cfg, params, ndm, record=rec)
    worst = 0.0
    for c in conv_specs(cfg):
        a = be.read_activation(h, c.name, (1, c.h_out, c.w_out, c.cout))
        r = rec.get(c.name + '+res', rec[c.name])
        worst = max(worst, _rel(a, r))
        assert _rel(a, r) < 1e-4, c.name
    # vote on identical maps, then the fused infer path end to end
    xyz = be.vote(h, hm, hm3, um, ndm, cfgs, coms)
    ref_same = pose.estimate_pose_mm(hm, hm3, um, ndm, cfgs, coms)
    np.testing.assert_array_equal(xyz, ref_same)                # the vote is bit-reproducible (vote.h::vote_exp == oracle/pose.py::exp_f32)
    h.call('dr_set_fusion', 1)
    xyz2 = be.infer(h, ndm, cfgs, coms)
    assert pose.mean_jnt_error(xyz2, g['xyz']) <= 0.1           # BASELINE.json tolerance
    assert np.abs(xyz2 - g['xyz']).max() < 0.05
    h.close()


def test_network_bf16_precision(be, case1):
    """dr_set_precision(DR_PREC_BF16) (BASELINE config 5's "bf16 MFMA conv path"): every k != 7 conv rounds both
    operands to bf16 on their way into the matrix cores.  Two bf16 evaluations with different fp32 summation orders
    decorrelate (a 1e-7 difference flips a rounding of 2^-9), so the network-level statement is about noise, not
    bits: per conv output and per head map the engine is (a) closer to the oracle's bf16-operand evaluation than that
    evaluation is to fp32, and (b) no further from fp32 than the oracle's bf16 evaluation is (x1.25) -- i.e. it
    carries the precision's own error and nothing else.  Kernel-level exactness is test_conv_bf16_matrix_core_variant.
    The vote runs in fp32 on whatever maps it is given (test_network_forward_and_vote_config1)."""
    from oracle import net
    from oracle.graph import conv_specs
    cfg, params, g = case1
    h = be.handle(cfg, 1)
    h.call('dr_set_precision', 1)
    h.load_params(params)
    h.call('dr_finalize_params', be.stream)
    ndm = be.norm_dm(h, g['dm'], g['com'])
    hm, hm3, um = be.forward_eval(h, ndm)
    rec16, rec32 = {}, {}
    o16 = net.forward_eval(cfg, params, ndm, record=rec16, conv_operands='bf16')
    o32 = net.forward_eval(cfg, params, ndm, record=rec32)
    l2 = lambda a, b: float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(np.asarray(b).ravel()) + 1e-12))

    def check(a, r16, r32, what):
        e_eng16, e_prec, e_eng32 = l2(a, r16), l2(r16, r32), l2(a, r32)
        assert e_eng16 <= e_prec + 1e-5, (what, e_eng16, e_prec)
        assert e_eng32 <= 1.25 * e_prec + 1e-5, (what, e_eng32, e_prec)
        return e_eng32

    worst = 0.0
    for c in conv_specs(cfg):
        a = be.read_activation(h, c.name, (1, c.h_out, c.w_out, c.cout))
        key = c.name + '+res' if c.name + '+res' in rec16 else c.name
        worst = max(worst, check(a, rec16[key], rec32[key], c.name))
    for k, a in (('hm_outs', hm), ('hm3_outs', hm3), ('um_outs', um)):
        check(a, o16[k][-1], o32[k][-1], k)
    assert 1e-3 < worst < 0.15, worst                      # it is bf16 (not fp32), and it is not garbage
    # switching back re-packs fp32 weights and reproduces the fp32 maps
    h.call('dr_set_precision', 0)
    with pytest.raises(Exception):
        be.forward_eval(h, ndm)                            # un-finalized by the precision change
    h.call('dr_finalize_params', be.stream)
    hm_b, _, _ = be.forward_eval(h, ndm)
    assert np.abs(hm_b[:, ::2, ::2] - g['hm']).max() < 2e-4
    h.close()
    ht = be.handle(cfg, 1, training=True)
    with pytest.raises(Exception):
        ht.call('dr_set_precision', 7)
    ht.close()


def test_vote_crafted_cases(be):
    """Planted peaks / exact ties / negative weights / out-of-range re-projection / all-background."""
    from oracle.graph import NetConfig
    g = golden('vote_cases.npz')
    hm, hm3, um = (g[k].astype(np.float32) for k in ('hm', 'hm3', 'um'))
    B, m, _, J = hm.shape
    ndm = np.repeat(np.repeat(g['tiny'], 4, axis=1), 4, axis=2).astype(np.float32)
    h = be.handle(NetConfig(1, 8, J), B)
    xyz = be.vote(h, hm, hm3, um, ndm, g['cfg'], g['com'])
    np.testing.assert_array_equal(xyz, g['xyz'])                # bit for bit: every operation of the vote is a fixed IEEE fp32 sequence on both sides
    assert np.isfinite(xyz).all()
    # exact-tie sample: the five candidates are the first five pixels in row-major order
    assert g['idx'][4].tolist() == [[0, 1, 2, 3, 4]] * J
    h.close()


def test_vote_seeded_random_maps(be):
    """The vote kernel on seeded random maps (smooth bumps + noise, a third of the pixels background, cameras and centres of mass
    in the range of the synthetic crops) against the oracle's _xyz_estimation: top-5 selection, re-projection weights,
    4x4x4 start cell and ten mean-shift iterations, in mm."""
    from oracle.graph import NetConfig
    from oracle import pose
    rng = np.random.default_rng(314159)
    B, m, J = 4, 32, 7
    yy, xx = np.mgrid[0:m, 0:m].astype(np.float32)
    for rep in range(3):
        hm = 0.05 * rng.standard_normal((B, m, m, J)).astype(np.float32)
        hm3 = np.abs(0.05 * rng.standard_normal((B, m, m, J))).astype(np.float32)
        for b in range(B):
            for j in range(J):
                for _ in range(int(rng.integers(1, 4))):                      # a few bumps per joint map
                    cy, cx, a = rng.uniform(3, m - 3), rng.uniform(3, m - 3), rng.uniform(0.3, 1.0)
                    bump = (a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * rng.uniform(1.0, 3.0) ** 2))).astype(np.float32)
                    hm[b, :, :, j] += bump
                    hm3[b, :, :, j] += bump * np.float32(rng.uniform(0.5, 1.0))
        um = (0.3 * rng.standard_normal((B, m, m, 3 * J))).astype(np.float32)
        tiny = rng.uniform(-0.4, 0.9, (B, m, m, 1)).astype(np.float32)
        tiny[rng.random((B, m, m, 1)) < 0.33] = -1.0                          # background
        ndm = np.repeat(np.repeat(tiny, 4, axis=1), 4, axis=2).astype(np.float32)
        cfg = np.stack([[rng.uniform(380, 720), rng.uniform(380, 720), 64 + rng.normal(0, 3), 64 + rng.normal(0, 3), 128, 128]
                        for _ in range(B)]).astype(np.float32)
        com = np.stack([[rng.normal(0, 40), rng.normal(0, 40), rng.uniform(250, 900)] for _ in range(B)]).astype(np.float32)
        want = pose.estimate_pose_mm(hm, hm3, um, ndm, cfg, com)
        h = be.handle(NetConfig(1, 8, J), B)
        xyz = be.vote(h, hm, hm3, um, ndm, cfg, com)
        h.close()
        assert np.isfinite(xyz).all()
        np.testing.assert_array_equal(xyz, want, err_msg='rep %d' % rep)     # (was: 5e-3 mm while exp() differed by ulps)


def test_abi_error_behaviour(be):
    """Argument checking mirrors the reference's error behaviour (ValueError on unknown input size,
    um_v1.py:106-107) plus the C-ABI contract of SURVEY 8b."""
    from densereg_amd import _lib
    from oracle.graph import NetConfig
    with pytest.raises(_lib.DenseRegError) as e:
        _lib.Handle(be.lib, 1, 8, 2, 100, 3, 1, 0, False)
    assert e.value.code == -2 and 'unknown input depth map shape' in str(e.value)
    with pytest.raises(_lib.DenseRegError):
        _lib.Handle(be.lib, 0, 8, 2, 128, 3, 1, 0, False)
    h = be.handle(NetConfig(1, 8, 2), 1)
    x = be.empty((2, 128, 128, 1))
    with pytest.raises(_lib.DenseRegError) as e:            # forward before finalize
        h.call('dr_forward_eval', 1, be.ptr(x), None, None, None, be.stream)
    assert e.value.code == -3
    h.call('dr_finalize_params', be.stream)
    with pytest.raises(_lib.DenseRegError) as e:            # B > max_batch
        h.call('dr_forward_eval', 2, be.ptr(x), None, None, None, be.stream)
    assert e.value.code == -1
    with pytest.raises(_lib.DenseRegError):
        h.call('dr_load_param', b'no/such/var', x.ctypes.data if hasattr(x, 'ctypes') else 1, 1)
    bad = np.zeros(3, np.float32)
    with pytest.raises(_lib.DenseRegError):                 # wrong element count
        h.call('dr_load_param', b'Conv/weights', bad.ctypes.data, 3)
    assert h.lib.dr_forward_eval(h._h, 1, None, None, None, None, None) == -1      # null input
    h.close()
