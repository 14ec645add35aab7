"""GPU parity of the whole trunk against the golden vectors generated from the reference
(tests/golden, oracle/gen_golden.py) and against the CPU oracle's intermediate activations.
Contract (BASELINE.json north_star): max abs logit error <= 1e-3 vs the reference fp32 CPU path."""
import os

import numpy as np
import pytest
import torch

from oracle import resnet_oracle as R
from oracle import weights as W

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3          # BASELINE.json north_star: "within 1e-3 on the output logits"
TAPS = ['pool'] + ['layer%d.%d' % (l, b) for l in (1, 2, 3, 4) for b in (0, 1)]


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def sd():
    return W.make_resnet18_state_dict(11)


@pytest.mark.parametrize('planes,tol', [(2, 2e-4), (3, 2e-4)])     # mx (fp6 cross terms) measures <= 7e-5 here: 3x headroom, not 8x
def test_taps_vs_oracle_64(dev, sd, planes, tol):
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    u8 = W.make_u8_patches(12, (2, 16, 3, 64, 64)).reshape(-1, 3, 64, 64)[:6]
    x = R.normalize_u8(u8)
    taps = {}
    with torch.no_grad():
        R.trunk(sd, x, taps)
    eng = TrunkEngine(sd, dev, planes=planes)
    report = []
    for i, name in enumerate(TAPS):
        got = eng.forward_f32(x.to(dev), tap=i).cpu()
        ref = taps[name]
        assert got.shape == ref.shape, name
        err = float((got - ref).abs().max() / ref.abs().max())
        report.append((name, err))
    print('planes=%d tap errors (rel to max):' % planes, report)
    for name, err in report:
        assert err <= tol, report


def test_u8_slide_path_equals_f32_path(dev, sd):
    """Fused tile read + transform in the stem == gather + normalise on the host side.  With the table look-up
    arithmetic the two are bit-identical; the default INTEGER arithmetic (transform and BN folded into 24-bit fixed-point
    weights, i8 MFMA with exact i32 accumulation) agrees to fp32 rounding and is the MORE accurate of the two against the
    fp32 oracle."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    rng = np.random.default_rng(7)
    slide = rng.integers(0, 256, (200, 260, 3), dtype=np.uint8)
    xy = np.array([[0, 0], [100, 50], [260 - 64, 200 - 64], [230, 170]], np.int32)     # last one hangs over the edge
    from oracle import wsi_oracle as WO
    tiles = np.stack([WO.read_tile(slide, int(x), int(y), 64, 64) for x, y in xy]).transpose(0, 3, 1, 2)
    x = R.normalize_u8(tiles)
    with torch.no_grad():
        taps = {}
        R.trunk(sd, x, taps)
    lib = native.load()
    for planes in (2, 3):
        eng = TrunkEngine(sd, dev, planes=planes)
        sl, xyd = torch.from_numpy(slide).to(dev), torch.from_numpy(xy).to(dev)
        b = eng.forward_f32(x.to(dev), feat=True)[0].clone()
        a = eng.forward_tiles(sl, xyd, 64, 64, feat=True, logits=False)[0].clone()
        try:
            native.check(lib.wsi_stem_set_mode(2, 32), 'stem mode')
            a_lut = eng.forward_tiles(sl, xyd, 64, 64, feat=True, logits=False)[0].clone()
        finally:
            lib.wsi_stem_set_mode(1, 64)
        if planes == 2:
            assert torch.equal(a_lut, b)
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= (2e-5 if planes == 2 else 1e-3) * scale
        # stem output (tap 0) straight against the oracle's pooled map: exact-integer path vs table path
        ref = taps['pool']
        p_f32 = eng.forward_f32(x.to(dev), tap=0).cpu()
        e_f32 = float((p_f32 - ref).abs().max() / ref.abs().max())
        p_u8 = eng.forward_tiles(sl, xyd, 64, 64, logits=False, tap=0).cpu()
        e_u8 = float((p_u8 - ref).abs().max() / ref.abs().max())
        print('planes=%d stem tap rel err vs oracle: f32 input path %.2e, u8 exact-integer path %.2e' % (planes, e_f32, e_u8))
        assert e_f32 <= (2e-5 if planes == 2 else 2e-4) and e_u8 <= (2e-5 if planes == 2 else 2e-4)


def test_integer_stem_extreme_pixels(dev, sd):
    """Saturated (255), black (0) and checkerboard 0/255 tiles, plus tiles leaving the slide on every side: the integer stem
    (bytes x - 128 = -128 ... 127, inside byte, exact zero padding) against the fp32 oracle's pooled stem output."""
    from oracle import wsi_oracle as WO
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    slide = np.zeros((192, 192, 3), np.uint8)
    slide[:64] = 255
    slide[64:128, :, 0] = 255
    yy, xx = np.mgrid[:64, :192]
    slide[128:] = (((yy + xx) % 2) * 255)[..., None]
    xy = np.array([[0, 0], [64, 64], [128, 128], [-20, -30], [150, 160], [-40, 100], [100, -40]], np.int32)
    tiles = np.stack([WO.read_tile(slide, int(x), int(y), 64, 64) for x, y in xy]).transpose(0, 3, 1, 2)
    with torch.no_grad():
        taps = {}
        R.trunk(sd, R.normalize_u8(tiles), taps)
    ref = taps['pool']
    for planes, tol in ((2, 2e-5), (3, 2e-4)):
        eng = TrunkEngine(sd, dev, planes=planes)
        got = eng.forward_tiles(torch.from_numpy(slide).to(dev), torch.from_numpy(xy).to(dev), 64, 64, logits=False, tap=0).cpu()
        err = float((got - ref).abs().max() / ref.abs().max())
        print('integer stem, extreme pixels, planes=%d: rel err %.2e' % (planes, err))
        assert err <= tol


def _bag(dev, sd, name, planes, golden_dir):
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    g = np.load(os.path.join(golden_dir, name))
    shape = tuple(int(v) for v in g['input_shape'])
    B, P = shape[:2]
    u8 = W.make_u8_patches(int(g['input_seed']), shape).reshape(-1, *shape[2:])      # image index b*P + p
    eng = TrunkEngine(sd, dev, planes=planes, head=(sd['fc0.weight'], sd['fc0.bias']))
    feat, logits, _ = eng.forward_f32(R.normalize_u8(u8).to(dev), feat=True, logits=True)
    singles = logits.view(B, P, 4).transpose(0, 1).reshape(P * B, 4)                  # row = p*B + b
    h = eng.linear(feat.view(B, P * 512), sd['fc.0.weight'], sd['fc.0.bias'], relu=True)
    ens = eng.linear(h, sd['fc.2.weight'], sd['fc.2.bias'])
    return (float(np.abs(singles.cpu().numpy() - g['singles']).max()), float(np.abs(ens.cpu().numpy() - g['ensemble']).max()))


def test_golden_bag64(dev, sd, golden_dir):
    e1, e2 = _bag(dev, sd, 'resnet18_bag64.npz', 2, golden_dir)
    print('bag64 parity-mode max abs err: singles %.2e ensemble %.2e' % (e1, e2))
    assert e1 <= LOGIT_TOL and e2 <= LOGIT_TOL


def test_golden_cfg1_256(dev, sd, golden_dir):
    e1, e2 = _bag(dev, sd, 'resnet18_cfg1_256.npz', 2, golden_dir)
    print('cfg1 256x256 parity-mode max abs err: singles %.2e ensemble %.2e' % (e1, e2))
    assert e1 <= LOGIT_TOL and e2 <= LOGIT_TOL


def test_golden_mode3_fp16_mx(dev, sd, golden_dir):
    """Precision mode 3 (fp16 main pass + MX-fp6 cross terms) against the same reference goldens and the same contract."""
    for name in ('resnet18_bag64.npz', 'resnet18_cfg1_256.npz'):
        e1, e2 = _bag(dev, sd, name, 3, golden_dir)
        print('%s mode-3 max abs err: singles %.2e ensemble %.2e' % (name, e1, e2))
        assert e1 <= LOGIT_TOL and e2 <= LOGIT_TOL


def test_speed_mode_error_is_reported_not_claimed(dev, sd, golden_dir):
    """Single-pass bf16 does NOT meet 1e-3; keep it honest and bounded."""
    e1, e2 = _bag(dev, sd, 'resnet18_cfg1_256.npz', 1, golden_dir)
    print('cfg1 256x256 speed-mode (single-pass bf16) max abs err: singles %.2e ensemble %.2e' % (e1, e2))
    assert e1 <= 0.15 and e2 <= 0.15


def test_golden_tile_logits_256(dev, sd, golden_dir):
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    g = np.load(os.path.join(golden_dir, 'tile_logits_256.npz'))
    cls = W.make_head_state_dict(int(g['cls_seed']), 'classifier')
    u8 = W.make_u8_patches(int(g['input_seed']), tuple(int(v) for v in g['input_shape']))
    eng = TrunkEngine(sd, dev, planes=2, head=(cls['fc.0.weight'], cls['fc.0.bias']))
    # feed the tiles as a vertical strip "slide" through the fused u8 path
    strip = np.ascontiguousarray(u8.transpose(0, 2, 3, 1).reshape(-1, 256, 3))
    xy = np.stack((np.zeros(8, np.int32), np.arange(8, dtype=np.int32) * 256), 1)
    _, logits, fmap = eng.forward_tiles(torch.from_numpy(strip).to(dev), torch.from_numpy(xy).to(dev), 256, 256,
                                        logits=True, fmap=True)
    err = float(np.abs(logits.cpu().numpy() - g['logits']).max())
    ferr = float(np.abs(fmap.cpu().numpy()[:, ::16] - g['fmap_sub']).max())
    print('tile logits max abs err %.2e, feature map err %.2e' % (err, ferr))
    assert err <= LOGIT_TOL and ferr <= 1e-3


@pytest.mark.parametrize('planes', [2, 3])
@pytest.mark.parametrize('shape', [(1, 32, 32), (1, 64, 64), (3, 96, 160), (2, 160, 64), (1, 512, 512), (5, 256, 256)])
def test_edge_patch_shapes_vs_oracle(dev, sd, shape, planes):
    """Smallest / non-square / large patches and ragged batch sizes through every stage (the 1x1 maps of
    32x32 patches, the gather fallback of 512x512 ones) against the CPU oracle."""
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    n, h, w = shape
    u8 = W.make_u8_patches(100 + h + w, (n, 3, h, w))
    x = R.normalize_u8(u8)
    eng = TrunkEngine(sd, dev, planes=planes, head=(sd['fc0.weight'], sd['fc0.bias']), max_batch=4)
    feat, logits, fmap = eng.forward_f32(x.to(dev), feat=True, logits=True, fmap=True)
    with torch.no_grad():
        rf = R.trunk(sd, x)
        rfeat = torch.flatten(torch.nn.functional.adaptive_avg_pool2d(rf, 1), 1)
        rlog = torch.nn.functional.linear(rfeat, sd['fc0.weight'], sd['fc0.bias'])
    assert fmap.shape == rf.shape
    assert float((fmap.cpu() - rf).abs().max() / rf.abs().max()) <= (2e-4 if planes == 2 else 1.5e-3)
    # pooled features: absolute 1e-3 in the bf16x3 mode; relative to their scale in mode 3 (the contract is on the logits)
    assert float((feat.cpu() - rfeat).abs().max()) <= (1e-3 if planes == 2 else 5e-4 * float(rfeat.abs().max()))
    assert float((logits.cpu() - rlog).abs().max()) <= LOGIT_TOL


def test_unfused_reference_kernels_agree(dev, sd):
    """The A/B forms kept in the library (two-kernel stem, gather stride-2 + separate downsample) give
    the same logits as the fused defaults."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    lib = native.load()
    u8 = W.make_u8_patches(77, (6, 3, 128, 128))
    x = R.normalize_u8(u8).to(dev)
    eng = TrunkEngine(sd, dev, planes=2, head=(sd['fc0.weight'], sd['fc0.bias']))
    base = eng.forward_f32(x, logits=True)[1].clone()
    try:
        native.check(lib.wsi_stem_set_mode(0, 32), 'stem mode')
        native.check(lib.wsi_conv_set_mode(0), 'conv mode')
        alt = eng.forward_f32(x, logits=True)[1].clone()
    finally:
        lib.wsi_stem_set_mode(1, 64)
        lib.wsi_conv_set_mode(1)
    assert float((alt - base).abs().max()) <= 1e-4


def test_integer_stem_forms_bit_identical(dev, sd):
    """The two launch forms of the integer stem - one strip per workgroup with the digit planes in registers
    (wsi_stem_set_mode(3, rows)) and two strips sharing them in LDS (default) - run the same arithmetic in the same order:
    bit-identical pooled outputs, including an odd number of strips (a half-filled last workgroup) and short segments."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    lib = native.load()
    rng = np.random.default_rng(17)
    slide = torch.from_numpy(rng.integers(0, 256, (700, 900, 3), dtype=np.uint8)).to(dev)
    for planes in (2, 3):
        eng = TrunkEngine(sd, dev, planes=planes)
        for tile, n in ((256, 3), (64, 5), (192, 1)):                   # 5 strips x 2 segments x 3; 1 strip x 1 x 5; 4 strips x 2 segments
            xy = torch.from_numpy(rng.integers(-30, 600, (n, 2)).astype(np.int32)).to(dev)
            for rows in (32, 7):
                try:
                    native.check(lib.wsi_stem_set_mode(1, rows), 'stem mode')
                    a = eng.forward_tiles(slide, xy, tile, tile, logits=False, tap=0).clone()
                    native.check(lib.wsi_stem_set_mode(3, rows), 'stem mode')
                    b = eng.forward_tiles(slide, xy, tile, tile, logits=False, tap=0).clone()
                finally:
                    lib.wsi_stem_set_mode(1, 64)
                assert torch.equal(a, b), (planes, tile, n, rows)


def test_ab_switches_agree(dev, sd):
    """Every A/B route of wsi_conv_set_mode (phase-slab vs wide stride-2 kernel, slab3 vs wide stride-1 kernel, XCD
    orders, 128-pixel stride-2 tiles) gives the same logits up to summation order."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    lib = native.load()
    u8 = W.make_u8_patches(78, (5, 3, 128, 128))
    x = R.normalize_u8(u8).to(dev)
    for planes, tol in ((3, 2e-4), (2, 2e-5)):
        eng = TrunkEngine(sd, dev, planes=planes, head=(sd['fc0.weight'], sd['fc0.bias']))
        base = eng.forward_f32(x, logits=True)[1].clone()
        try:
            # (+2048: the strided blocks' downsample as its own tensor + residual instead of folded into the second conv; +4096:
            #  layer-1 kernel without paired-tile addressing - that one must not change a bit)
            for mode in (3, 1 + 8, 1 + 16, 1 + 32, 1 + 128, 1 + 256, 1 + 512, 3 + 32 + 128 + 256, 1 + 2048, 1 + 2048 + 4096, 1 + 16384, 1 + 32768):
                native.check(lib.wsi_conv_set_mode(mode), 'conv mode')
                alt = eng.forward_f32(x, logits=True)[1].clone()
                err = float((alt - base).abs().max())
                assert err <= tol, (planes, mode, err)
        finally:
            lib.wsi_conv_set_mode(1)


def test_forward_is_bit_deterministic(dev, sd):
    """No atomics or order-dependent sums in the trunk: repeated runs of the same batch are bit-identical (both split modes)."""
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    g = torch.Generator(device=dev).manual_seed(5)
    slide = torch.randint(0, 256, (256 * 3, 256 * 5, 3), dtype=torch.uint8, device=dev, generator=g)
    xy = torch.tensor([[256 * (i % 5), 256 * (i // 5)] for i in range(15)], dtype=torch.int32, device=dev)
    for planes in (3, 2):
        eng = TrunkEngine(sd, dev, planes=planes, head=(sd['fc0.weight'], sd['fc0.bias']), max_batch=15)
        ref = eng.forward_tiles(slide, xy, 256, 256, logits=True)[1].clone()
        assert bool(torch.isfinite(ref).all())
        for _ in range(6):
            assert torch.equal(ref, eng.forward_tiles(slide, xy, 256, 256, logits=True)[1])


@pytest.mark.parametrize('planes', [1, 2, 3])
def test_trunk_set_chunks_equals_unchunked(dev, sd, planes):
    """wsi_trunk_set_chunks (sub-batches of the stem / layer-1 stages) must not change a single bit in any precision
    mode (r01 bug: the per-image offset used 6 B/channel for the mx format, which has 4).  Since r03 a layer-1 sub-batch
    writes its images' slice of the phase-split hand-over to the wide stride-2 kernel too (before, a layer-1 chunk
    smaller than the batch fell back to the unsplit stride-2 kernel, which refuses tensors >= 4 GB), so every setting
    runs the same kernels on every stage."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    lib = native.load()
    u8 = W.make_u8_patches(21, (1, 7, 3, 64, 64)).reshape(-1, 3, 64, 64)
    strip = np.ascontiguousarray(u8.transpose(0, 2, 3, 1).reshape(-1, 64, 3))
    xy = np.stack((np.zeros(7, np.int32), np.arange(7, dtype=np.int32) * 64), 1)
    sl, xyd = torch.from_numpy(strip).to(dev), torch.from_numpy(xy).to(dev)
    eng = TrunkEngine(sd, dev, planes=planes, head=(sd['fc0.weight'], sd['fc0.bias']))
    x = R.normalize_u8(u8).to(dev)

    def run():
        return ([t.clone() for t in eng.forward_tiles(sl, xyd, 64, 64, feat=True, logits=True, fmap=True)] +
                [t.clone() for t in eng.forward_f32(x, feat=True, logits=True, fmap=True)])
    base = run()
    try:
        for cs, c1 in ((2, 4), (1, 1), (3, 0), (0, 2), (7, 7), (1, 7)):
            native.check(lib.wsi_trunk_set_chunks(cs, c1), 'wsi_trunk_set_chunks')
            for a, b in zip(run(), base):
                assert torch.equal(a, b), (planes, cs, c1)
    finally:
        lib.wsi_trunk_set_chunks(0, 0)
    assert lib.wsi_trunk_set_chunks(2, 3) != 0                    # layer1 chunk must be a multiple of the stem chunk


def test_paired_tile_addressing_is_bit_identical(dev, sd):
    """conv3x3s1_slab3_kernel<.., PAIR> (odd pixel tiles read at the even tile's LDS addresses + 4096 on 64-wide maps) against
    the same kernel with every address computed: identical bits on 256x256 tiles (layer 1 is the only 64-wide stage), with and
    without the residual; and 64x64 tiles (16-wide layer 1: PAIR not applicable) still run."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    lib = native.load()
    g = torch.Generator(device=dev).manual_seed(9)
    slide = torch.randint(0, 256, (256 * 2, 256 * 3, 3), dtype=torch.uint8, device=dev, generator=g)
    xy = torch.tensor([[256 * (i % 3), 256 * (i // 3)] for i in range(6)], dtype=torch.int32, device=dev)
    eng = TrunkEngine(sd, dev, planes=3, head=(sd['fc0.weight'], sd['fc0.bias']), max_batch=6)
    base = [t.clone() for t in eng.forward_tiles(slide, xy, 256, 256, feat=True, logits=True, fmap=True)]
    try:
        native.check(lib.wsi_conv_set_mode(1 + 4096), 'conv mode')
        alt = eng.forward_tiles(slide, xy, 256, 256, feat=True, logits=True, fmap=True)
        for a, b in zip(alt, base):
            assert torch.equal(a, b)
    finally:
        lib.wsi_conv_set_mode(1)
    small = torch.randint(0, 256, (64, 64 * 5, 3), dtype=torch.uint8, device=dev, generator=g)
    xy5 = torch.tensor([[64 * i, 0] for i in range(5)], dtype=torch.int32, device=dev)
    assert bool(torch.isfinite(eng.forward_tiles(small, xy5, 64, 64, logits=True)[1]).all())


def test_96_byte_layer1_lines_are_bit_identical(dev, sd):
    """r03: a full mx trunk run keeps the stem output and the layer-1 tensors in 96-byte lines (no hi6 plane in memory; the
    layer-1 kernel rebuilds it in LDS from the fp16 plane with one v_cvt_scalef32_pk32_fp6_f16 per line) - identical bits to the
    128-byte route (wsi_conv_set_mode + 16384) for the u8 and the f32 input paths, on 256x256 tiles (64-wide layer 1, paired
    tiles), 64x64 crops (16-wide) and 128x192 tiles; and a workspace that alternates between full runs (96-byte lines in the
    stage-0 buffers) and tap runs (128-byte lines in the SAME buffers, whose pad bytes sit elsewhere) stays exact."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    lib = native.load()
    g = torch.Generator(device=dev).manual_seed(19)
    eng = TrunkEngine(sd, dev, planes=3, head=(sd['fc0.weight'], sd['fc0.bias']), max_batch=7)
    for th, tw, n in ((256, 256, 7), (64, 64, 5), (128, 192, 3)):
        slide = torch.randint(0, 256, (th * 2, tw * 4, 3), dtype=torch.uint8, device=dev, generator=g)
        xy = torch.tensor([[tw * (i % 4), th * (i // 4)] for i in range(n)], dtype=torch.int32, device=dev)
        x = torch.randn(n, 3, th, tw, device=dev, generator=g)

        def run():
            return ([t.clone() for t in eng.forward_tiles(slide, xy, th, tw, feat=True, logits=True, fmap=True)] +
                    [t.clone() for t in eng.forward_f32(x, feat=True, logits=True, fmap=True)])
        new = run()
        tap1 = eng.forward_tiles(slide, xy, th, tw, logits=False, tap=1).clone()      # 128-byte lines in the buffers the full run left in 96
        again = run()                                                              # ... and back
        try:
            native.check(lib.wsi_conv_set_mode(1 + 16384), 'conv mode')
            old = run()
            tap1_old = eng.forward_tiles(slide, xy, th, tw, logits=False, tap=1).clone()
        finally:
            lib.wsi_conv_set_mode(1)
        for a, b, c in zip(new, old, again):
            assert torch.equal(a, b) and torch.equal(a, c), (th, tw)
        assert torch.equal(tap1, tap1_old)


def test_256_cout_stride2_workgroups_are_bit_identical(dev, sd):
    """conv3x3s2_wide_kernel<.., NT = 4> (256 px x 256 couts per workgroup on the layer-3 / layer-4 entries, one step per barrier,
    two 32 KB weight stages) sums every output in the same order as the 128-cout form (wsi_conv_set_mode + 32768): identical bits,
    on 256x256 tiles (dense 16x16 / 8x8 output maps) and 64x64 crops (4x4 / 2x2: the non-dense form)."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    lib = native.load()
    g = torch.Generator(device=dev).manual_seed(23)
    eng = TrunkEngine(sd, dev, planes=3, head=(sd['fc0.weight'], sd['fc0.bias']), max_batch=9)
    for t, n in ((256, 9), (64, 7)):
        slide = torch.randint(0, 256, (t * 3, t * 3, 3), dtype=torch.uint8, device=dev, generator=g)
        xy = torch.tensor([[t * (i % 3), t * (i // 3)] for i in range(n)], dtype=torch.int32, device=dev)
        new = [v.clone() for v in eng.forward_tiles(slide, xy, t, t, feat=True, logits=True, fmap=True)]
        try:
            native.check(lib.wsi_conv_set_mode(1 + 32768), 'conv mode')
            old = eng.forward_tiles(slide, xy, t, t, feat=True, logits=True, fmap=True)
            for a, b in zip(new, old):
                assert torch.equal(a, b), t
        finally:
            lib.wsi_conv_set_mode(1)


def test_eight_pixel_slab_rows_are_bit_identical(dev, sd):
    """conv3x3s1_wide_kernel<.., D8> (r05: on 8 x 8 maps the LDS slab keeps rows of eight pixels - no pad column - and edge lanes
    read zero pixels, which removes the bank conflicts of the 9-pixel pitch) multiplies the same operands in the same order as the
    padded-flat slab image (wsi_conv_set_mode + 131072): identical bits in all three precision modes, on batches that fill whole
    256-pixel tiles (8 tiles = 2 tiles of four images), a ragged last tile (9, 3, 1 images) and through the extra K segment of the
    strided block (layer4.0.conv2 carries the folded 1x1 downsample in mx)."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    lib = native.load()
    g = torch.Generator(device=dev).manual_seed(29)
    for planes in (3, 2, 1):
        eng = TrunkEngine(sd, dev, planes=planes, head=(sd['fc0.weight'], sd['fc0.bias']), max_batch=16)
        for n in (8, 9, 3, 1):
            slide = torch.randint(0, 256, (256 * 3, 256 * 3, 3), dtype=torch.uint8, device=dev, generator=g)
            xy = torch.tensor([[256 * (i % 3), 256 * (i // 3)] for i in range(n)], dtype=torch.int32, device=dev)
            new = [v.clone() for v in eng.forward_tiles(slide, xy, 256, 256, feat=True, logits=True, fmap=True)]
            try:
                native.check(lib.wsi_conv_set_mode(1 + 131072), 'conv mode')
                old = eng.forward_tiles(slide, xy, 256, 256, feat=True, logits=True, fmap=True)
                for a, b in zip(new, old):
                    assert torch.equal(a, b), (planes, n)
            finally:
                lib.wsi_conv_set_mode(1)


def test_persistent_layer1_kernel_is_bit_identical(dev, sd):
    """conv3x3s1_l1p_kernel (r05: one persistent workgroup per CU, a producer wave issues every LDS-DMA one unit ahead, residual tiles
    staged in a 96-byte pitch) multiplies and sums exactly as conv3x3s1_rows_kernel (wsi_conv_set_mode + 1048576): identical bits on
    batches that give every workgroup several tiles (40 tiles = 640 layer-1 tiles on 256 workgroups), fewer tiles than workgroups
    (1, 3, 9 tiles) and a ragged split over the eight XCD ranges (17)."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    lib = native.load()
    g = torch.Generator(device=dev).manual_seed(31)
    eng = TrunkEngine(sd, dev, planes=3, head=(sd['fc0.weight'], sd['fc0.bias']), max_batch=64)
    slide = torch.randint(0, 256, (256 * 7, 256 * 6, 3), dtype=torch.uint8, device=dev, generator=g)
    for n in (40, 17, 9, 3, 1):
        xy = torch.tensor([[256 * (i % 6), 256 * (i // 6)] for i in range(n)], dtype=torch.int32, device=dev)
        new = [v.clone() for v in eng.forward_tiles(slide, xy, 256, 256, feat=True, logits=True, fmap=True)]
        try:
            native.check(lib.wsi_conv_set_mode(1 + 1048576), 'conv mode')
            old = eng.forward_tiles(slide, xy, 256, 256, feat=True, logits=True, fmap=True)
            for a, b in zip(new, old):
                assert torch.equal(a, b), n
        finally:
            lib.wsi_conv_set_mode(1)
