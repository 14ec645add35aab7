"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes) and checked
against the CPU oracle (torch fp32 ops on the CPU / numpy restatements)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_PARITY = 2e-5      # relative to the tensor's max magnitude, fp16 hi + fp16 lo pair (3 MFMA passes; r01-r04 bf16 pair: 1e-4)
TOL_SPEED = 3e-2       # single-pass bf16
TOL_MX = 1e-4          # fp16 main pass + MX-fp6 cross terms (mode 3), single layer (measured <= 3e-5)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    return torch.device('cuda:0')


def _rel_err(got, ref):
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))


def test_pf_roundtrip(dev):
    from wsi_segmentation_pipeline_amd import engine as E
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 128, 5, 7, generator=g)
    for planes, tol in ((2, 2 ** -21), (1, 2 ** -8), (3, 2 ** -13)):      # planes 2: fp16 pair
        buf = E.pf_pack(x.to(dev), planes)
        y = E.pf_unpack(buf, 3, 128, 5, 7, planes).cpu()
        assert _rel_err(y, x) <= tol
    # pad positions stay zero: total sum of the raw buffer equals the sum over real pixels only
    buf = E.pf_pack(torch.ones(2, 64, 4, 4, device=dev), 1)
    assert int((buf.view(torch.int16) != 0).sum()) == 2 * 64 * 16


def _conv_case(dev, n, cin, cout, h, w, stride, ksize, resid, relu, planes, seed):
    from wsi_segmentation_pipeline_amd import engine as E
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g).abs_()
    wt = torch.randn(cout, cin, ksize, ksize, generator=g) * (2.0 / (cin * ksize * ksize)) ** 0.5
    bn = (torch.rand(cout, generator=g) * 0.5 + 0.75, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) * 0.5 + 0.75)
    ho, wo = h // stride, w // stride
    r = torch.randn(n, cout, ho, wo, generator=g) if resid else None
    ref = F.conv2d(x, wt, None, stride, ksize // 2)
    ref = F.batch_norm(ref, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    if r is not None:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    wpk, bias = E.prepack_conv(wt, bn, planes, dev)
    xpf = E.pf_pack(x.to(dev), planes)
    rpf = E.pf_pack(r.to(dev), planes) if r is not None else None
    opf = E.conv_bn_act(xpf, n, h, w, cin, cout, wpk, bias, stride, ksize, rpf, relu, planes)
    got = E.pf_unpack(opf, n, cout, ho, wo, planes).cpu()
    torch.cuda.synchronize()
    # pad positions of the output buffer must still be zero (next layer's implicit padding)
    if planes == 3:          # pad pixels: compare whole 128-byte lines of pad positions against zero via the hi plane mask
        real = E.pf_pack(torch.ones_like(got).to(dev), 3).view(-1, 128)[:, :64].ne(0).any(1)
        assert not bool(opf.view(-1, 128)[~real].ne(0).any()), 'kernel wrote to a pad position'
    else:
        real = E.pf_pack(torch.full_like(got, 1.0 + 2.0 ** -12).to(dev), planes).view(torch.int16) != 0
        assert not bool((opf.view(torch.int16)[~real] != 0).any()), 'kernel wrote to a pad position'
    return _rel_err(got, ref)


def test_parity_weight_scales_cover_any_magnitude(dev):
    """Mode 2 (fp16 pair, r05): wsi_prepack_conv multiplies every output channel's folded weights by a power of two that puts the
    channel's largest magnitude into [2^13, 2^14) and the epilogue divides the accumulators by it - so channels whose weights sit deep in
    fp16's subnormals (6e-5 x N(0, 1): without the scale their lo parts would all be subnormal, ~1e-3 relative) or far above 1 (gains 1e-3 ...
    1e3 across the channels of one conv; activations of ordinary size) keep the pair's precision, and an all-zero channel stays zero."""
    from wsi_segmentation_pipeline_amd import engine as E
    g = torch.Generator().manual_seed(17)
    n, cin, cout, h, w = 2, 64, 64, 16, 16
    x = torch.randn(n, cin, h, w, generator=g).abs_()
    gain = torch.logspace(3, -3, cout).view(-1, 1, 1, 1)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5 * gain
    wt[5] = 0
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    wpk, bias = E.prepack_conv(wt, None, 2, dev)
    out = E.pf_unpack(E.conv_bn_act(E.pf_pack(x.to(dev), 2), n, h, w, cin, cout, wpk, bias, 1, 3, None, False, 2), n, cout, h, w, 2).cpu().double()
    scale = ref.abs().amax((0, 2, 3)).clamp_min(1e-300)                    # per output channel: each has its own magnitude
    rel = ((out - ref).abs().amax((0, 2, 3)) / scale)
    rel[5] = out[:, 5].abs().max()                                         # the zero channel: exactly zero
    print('per-channel rel err: max %.2e (channel %d)' % (float(rel.max()), int(rel.argmax())))
    assert float(rel.max()) <= 2e-5                                        # (outputs above 65504 would be clamped: none here)
    assert float(ref.abs().max()) < 65504


CONV_S1 = [  # n, cin, cout, h, w
    (3, 64, 64, 16, 16), (2, 64, 64, 64, 64), (5, 128, 128, 8, 8), (2, 128, 128, 32, 32),
    (3, 256, 256, 4, 4), (2, 256, 256, 16, 16), (7, 512, 512, 2, 2), (3, 512, 512, 8, 8), (1, 64, 128, 8, 8),
    (2, 64, 64, 12, 256), (1, 64, 64, 5, 130), (1, 128, 64, 6, 200),        # maps wider than 128: the 512-pixel slab3 tiles (cfg 39)
    (3, 64, 64, 128, 64), (1, 64, 64, 6, 64), (2, 128, 64, 8, 64),          # 64-wide maps: the row-stacked kernel (cfg 40; H % 4 != 0 falls back to slab3)
]


@pytest.mark.parametrize('shape', CONV_S1)
def test_conv3x3_stride1(dev, shape):
    n, cin, cout, h, w = shape
    assert _conv_case(dev, n, cin, cout, h, w, 1, 3, True, True, 2, 1) <= TOL_PARITY
    assert _conv_case(dev, n, cin, cout, h, w, 1, 3, False, False, 2, 2) <= TOL_PARITY
    assert _conv_case(dev, n, cin, cout, h, w, 1, 3, True, True, 1, 3) <= TOL_SPEED
    e3 = (_conv_case(dev, n, cin, cout, h, w, 1, 3, True, True, 3, 1), _conv_case(dev, n, cin, cout, h, w, 1, 3, False, False, 3, 2))
    print('mode-3 conv rel err', shape, e3)
    assert max(e3) <= TOL_MX


@pytest.mark.parametrize('shape', [(2, 32, 32, 40, 72), (1, 128, 32, 20, 130), (3, 32, 32, 9, 300), (1, 64, 32, 33, 64)])
def test_conv3x3_32_channel_lines(dev, shape):
    """Split-precision tensors of 32 channels (one 128-byte line): the stride-1 3x3 kernel with 32-output-channel tiles (cfg 90,
    U-Net decoder levels 4-5); speed mode keeps 64-channel lines and refuses them."""
    n, cin, cout, h, w = shape
    assert _conv_case(dev, n, cin, cout, h, w, 1, 3, False, True, 2, 11) <= TOL_PARITY
    assert _conv_case(dev, n, cin, cout, h, w, 1, 3, False, True, 3, 12) <= TOL_MX
    from wsi_segmentation_pipeline_amd import native
    assert native.load().wsi_prepack_conv_bytes(cout, 32, 3, 1) == 0


CONV_S2 = [(3, 64, 128, 16, 16), (2, 64, 128, 64, 64), (3, 128, 256, 8, 8), (2, 256, 512, 16, 16), (5, 256, 512, 4, 4)]


@pytest.mark.parametrize('shape', CONV_S2)
def test_conv_stride2_and_downsample(dev, shape):
    n, cin, cout, h, w = shape
    assert _conv_case(dev, n, cin, cout, h, w, 2, 3, False, True, 2, 4) <= TOL_PARITY
    assert _conv_case(dev, n, cin, cout, h, w, 2, 1, False, False, 2, 5) <= TOL_PARITY
    assert _conv_case(dev, n, cin, cout, h, w, 2, 3, False, True, 1, 6) <= TOL_SPEED
    assert _conv_case(dev, n, cin, cout, h, w, 2, 1, False, False, 1, 7) <= TOL_SPEED
    assert _conv_case(dev, n, cin, cout, h, w, 2, 3, False, True, 3, 4) <= TOL_MX       # mode 3: slab form only


def test_mfma_orientation_asymmetric(dev):
    """A = identity-like weights with an asymmetric input catches a swapped C/D mapping."""
    from wsi_segmentation_pipeline_amd import engine as E
    n, c, h, w = 1, 64, 8, 8
    x = torch.arange(n * c * h * w, dtype=torch.float32).reshape(n, c, h, w) % 251
    wt = torch.zeros(c, c, 3, 3)
    for co in range(c):
        wt[co, (co * 7 + 3) % c, 1, 1] = 1.0           # a channel permutation at the centre tap
        wt[co, (co * 5 + 1) % c, 0, 2] = 0.5
    ref = F.conv2d(x, wt, None, 1, 1)
    wpk, bias = E.prepack_conv(wt, None, 2, dev)
    got = E.pf_unpack(E.conv_bn_act(E.pf_pack(x.to(dev), 2), n, h, w, c, c, wpk, bias, 1, 3, None, False, 2), n, c, h, w, 2)
    assert torch.equal(got.cpu(), ref)                  # small integers: exact


def test_linear(dev):
    from wsi_segmentation_pipeline_amd.engine import TrunkEngine
    from wsi_segmentation_pipeline_amd import native, engine as E
    import ctypes as C
    g = torch.Generator().manual_seed(3)
    x, w, b = torch.randn(11, 8192, generator=g), torch.randn(96, 8192, generator=g) * 0.01, torch.randn(96, generator=g)
    lib = native.load()
    y = torch.empty(11, 96, device=dev)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    native.check(lib.wsi_linear(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), 11, 8192, 96, 1,
                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'linear')
    ref = F.relu(F.linear(x, w, b))
    assert float((y.cpu() - ref).abs().max()) <= 1e-4


def test_avgpool_fc(dev):
    from wsi_segmentation_pipeline_amd import native, engine as E
    import ctypes as C
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5, 512, 8, 8, generator=g).abs_()
    w, b = torch.randn(4, 512, generator=g) * 0.05, torch.randn(4, generator=g)
    lib = native.load()
    feat, logit = torch.empty(5, 512, device=dev), torch.empty(5, 4, device=dev)
    wd, bd = w.to(dev), b.to(dev)
    native.check(lib.wsi_avgpool_fc(E.pf_pack(x.to(dev), 2).data_ptr(), 5, 8, 8, 512, wd.data_ptr(), bd.data_ptr(), 4,
                                    feat.data_ptr(), logit.data_ptr(), 2, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                 'avgpool_fc')
    rf = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    assert float((feat.cpu() - rf).abs().max()) <= 1e-5
    assert float((logit.cpu() - F.linear(rf, w, b)).abs().max()) <= 1e-4


def test_tile_gather_stitch_softmax(dev):
    from oracle import wsi_oracle as WO
    from oracle.resnet_oracle import normalize_u8, DATASET_MEAN, DATASET_STD
    from wsi_segmentation_pipeline_amd import engine as E
    rng = np.random.default_rng(5)
    slide = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    xy = np.array([[0, 0], [7, 13], [400 - 64, 300 - 64], [390, 290], [-5, -3]], np.int32)   # incl. out-of-slide reads
    lut = torch.from_numpy(E.normalize_lut(DATASET_MEAN, DATASET_STD)).to(dev)
    got = E.tile_gather(torch.from_numpy(slide).to(dev), torch.from_numpy(xy), 64, 64, lut).cpu()
    ref = torch.stack([normalize_u8(WO.read_tile(slide, int(x), int(y), 64, 64).transpose(2, 0, 1)[None])[0] for x, y in xy])
    assert torch.equal(got, ref)
    # stitch: overlapping footprints, clipping at the map border, exact float64 sums
    T, Cc, dy, dx, MH, MW = 200, 4, 8, 8, 61, 77
    logits = rng.standard_normal((T, Cc)).astype(np.float32)
    txy = np.stack((rng.integers(-3, MW, T), rng.integers(-3, MH, T)), 1).astype(np.int32)
    txy = np.maximum(txy, 0)                     # numpy slices with negative starts wrap; the reference never has them
    ref = np.zeros((Cc, MH, MW))
    for (x, y), p in zip(txy, logits):
        ref[:, y:y + dy, x:x + dx] += p[:, None, None].astype(np.float64)
    pred = torch.zeros((Cc, MH, MW), dtype=torch.float64, device=dev)
    E.stitch_add(pred, torch.from_numpy(logits).to(dev), torch.from_numpy(txy), dy, dx)
    assert np.array_equal(pred.cpu().numpy(), ref)
    # softmax / threshold / argmax / heat map
    mask = (rng.random((MH, MW)) > 0.3).astype(np.uint8)
    for th in ((0., 0., 0., 0.), (0.1, 0.3, 0.2, 0.25)):
        rc, rp = WO.threshold_probs(ref, th)
        cls, probs, heat = E.softmax_threshold_argmax(pred, th, torch.from_numpy(mask), 'cls')
        # probabilities: device f64 exp vs torch-CPU f64 exp may differ in the last bits (neither is correctly rounded)
        assert np.abs(probs.cpu().numpy() - rp).max() <= 4 * np.finfo(np.float64).eps
        # byte / index outputs are held to bit-exact: the COUNT of differing pixels is printed and must be 0
        rh = WO.tumorbed_heatmap(rp, mask, 'cls')
        n_cls = int((cls.cpu().numpy() != rc).sum())
        n_heat = int((heat.cpu().numpy() != rh).sum())
        print('thresholds %s: %d class pixels, %d heat pixels differ of %d' % (th, n_cls, n_heat, rc.size))
        assert n_cls == 0 and n_heat == 0


@pytest.mark.parametrize('shape', [(3, 64, 128, 16, 16), (2, 64, 128, 64, 64), (3, 128, 256, 32, 32), (5, 256, 512, 16, 16), (7, 256, 512, 4, 4)])
def test_fused_stride2_block_entry(dev, shape):
    """3x3 stride-2 conv (+ReLU) and the 1x1 stride-2 downsample in one launch == two separate convs."""
    import ctypes as C
    from wsi_segmentation_pipeline_amd import native, engine as E
    n, cin, cout, h, w = shape
    lib = native.load()
    for planes, tol in ((2, TOL_PARITY), (1, TOL_SPEED), (3, TOL_MX)):
        g = torch.Generator().manual_seed(9)
        x = torch.randn(n, cin, h, w, generator=g).abs_()
        w3 = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        w1 = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
        bn3 = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1,
               torch.rand(cout, generator=g) + 0.5)
        bn1 = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1,
               torch.rand(cout, generator=g) + 0.5)
        ref3 = F.relu(F.batch_norm(F.conv2d(x, w3, None, 2, 1), bn3[2], bn3[3], bn3[0], bn3[1], False, 0.0, 1e-5))
        ref1 = F.batch_norm(F.conv2d(x, w1, None, 2, 0), bn1[2], bn1[3], bn1[0], bn1[1], False, 0.0, 1e-5)
        wp3, b3 = E.prepack_conv(w3, bn3, planes, dev)
        wp1, b1 = E.prepack_conv(w1, bn1, planes, dev)
        xpf = E.pf_pack(x.to(dev), planes)
        o3, o1 = E.pf_zeros(n, cout, h // 2, w // 2, planes, dev), E.pf_zeros(n, cout, h // 2, w // 2, planes, dev)
        native.check(lib.wsi_conv3x3s2_ds_fused(xpf.data_ptr(), o3.data_ptr(), o1.data_ptr(), wp3.data_ptr(), b3.data_ptr(),
                                                wp1.data_ptr(), b1.data_ptr(), n, h, w, cin, cout, planes,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'fused s2')
        g3 = E.pf_unpack(o3, n, cout, h // 2, w // 2, planes).cpu()
        g1 = E.pf_unpack(o1, n, cout, h // 2, w // 2, planes).cpu()
        assert _rel_err(g3, ref3) <= tol and _rel_err(g1, ref1) <= tol
        if planes == 3:
            real = E.pf_pack(torch.ones_like(g3).to(dev), 3).view(-1, 128)[:, :64].ne(0).any(1)
            assert not bool(o3.view(-1, 128)[~real].ne(0).any()) and not bool(o1.view(-1, 128)[~real].ne(0).any())
        else:
            real = E.pf_pack(torch.full_like(g3, 1.0 + 2.0 ** -12).to(dev), planes).view(torch.int16) != 0
            assert not bool((o3.view(torch.int16)[~real] != 0).any()) and not bool((o1.view(torch.int16)[~real] != 0).any())


@pytest.mark.parametrize('shape', [(3, 64, 128, 16, 16), (2, 64, 128, 64, 64), (3, 128, 256, 32, 32), (5, 256, 512, 16, 16), (7, 256, 512, 4, 4), (1, 64, 128, 2, 2)])
def test_phase_split_stride2_block(dev, shape):
    """Stride-1 conv writing its output phase-split (wsi_conv3x3_bn_act_split) feeding the wide stride-2 kernel
    (wsi_conv3x3s2_ds_fused_split) == conv -> (stride-2 conv + ReLU, 1x1 stride-2 downsample) of the oracle."""
    import ctypes as C
    from wsi_segmentation_pipeline_amd import native, engine as E
    n, cin, cout, h, w = shape
    lib = native.load()
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for planes, tol in ((2, 2 * TOL_PARITY), (3, 2 * TOL_MX)):
        g = torch.Generator().manual_seed(13)
        x = torch.randn(n, cin, h, w, generator=g).abs_()
        w0 = torch.randn(cin, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        w3 = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        w1 = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
        mkbn = lambda c: (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1,
                          torch.rand(c, generator=g) + 0.5)
        bn0, bn3, bn1 = mkbn(cin), mkbn(cout), mkbn(cout)
        bnf = lambda t, b: F.batch_norm(t, b[2], b[3], b[0], b[1], False, 0.0, 1e-5)
        mid = F.relu(bnf(F.conv2d(x, w0, None, 1, 1), bn0) + x)
        ref3 = F.relu(bnf(F.conv2d(mid, w3, None, 2, 1), bn3))
        ref1 = bnf(F.conv2d(mid, w1, None, 2, 0), bn1)
        wp0, b0 = E.prepack_conv(w0, bn0, planes, dev)
        wp3, b3 = E.prepack_conv(w3, bn3, planes, dev)
        wp1, b1 = E.prepack_conv(w1, bn1, planes, dev)
        xpf = E.pf_pack(x.to(dev), planes)
        split = torch.zeros(lib.wsi_pf_split_bytes(n, h, w, cin, planes), dtype=torch.uint8, device=dev)
        native.check(lib.wsi_conv3x3_bn_act_split(xpf.data_ptr(), split.data_ptr(), xpf.data_ptr(), wp0.data_ptr(), b0.data_ptr(),
                                                  n, h, w, cin, cin, 1, planes, st()), 'conv split')
        # the four phase images unpack to the four sub-sampled grids of the intermediate
        per = lib.wsi_pf_bytes(n, h // 2, w // 2, cin, planes)
        for ph in range(4):
            img = E.pf_unpack(split[ph * per:(ph + 1) * per], n, cin, h // 2, w // 2, planes).cpu()
            assert _rel_err(img, mid[:, :, (ph >> 1)::2, (ph & 1)::2]) <= tol
        o3, o1 = E.pf_zeros(n, cout, h // 2, w // 2, planes, dev), E.pf_zeros(n, cout, h // 2, w // 2, planes, dev)
        native.check(lib.wsi_conv3x3s2_ds_fused_split(split.data_ptr(), o3.data_ptr(), o1.data_ptr(), wp3.data_ptr(), b3.data_ptr(),
                                                      wp1.data_ptr(), b1.data_ptr(), n, h, w, cin, cout, planes, st()), 'fused s2 split')
        g3 = E.pf_unpack(o3, n, cout, h // 2, w // 2, planes).cpu()
        g1 = E.pf_unpack(o1, n, cout, h // 2, w // 2, planes).cpu()
        e3, e1 = _rel_err(g3, ref3), _rel_err(g1, ref1)
        print('phase-split block', shape, 'planes', planes, 'rel err', e3, e1)
        assert e3 <= tol and e1 <= tol
        if planes == 3:
            real = E.pf_pack(torch.ones_like(g3).to(dev), 3).view(-1, 128)[:, :64].ne(0).any(1)
            assert not bool(o3.view(-1, 128)[~real].ne(0).any()) and not bool(o1.view(-1, 128)[~real].ne(0).any())


def test_phase_split_argument_errors(dev):
    """The phase-split entry points refuse what the wide stride-2 kernel cannot take (-22 = EINVAL), they never fault."""
    import ctypes as C
    from wsi_segmentation_pipeline_amd import native, engine as E
    lib = native.load()
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(3)
    w0 = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    w3 = torch.randn(128, 64, 3, 3, generator=g) * 0.05
    w1 = torch.randn(128, 64, 1, 1, generator=g) * 0.1
    for planes in (2, 3):
        wp0, b0 = E.prepack_conv(w0, None, planes, dev)
        wp3, b3 = E.prepack_conv(w3, None, planes, dev)
        wp1, b1 = E.prepack_conv(w1, None, planes, dev)
        buf = torch.zeros(1 << 22, dtype=torch.uint8, device=dev)
        out = torch.zeros(1 << 22, dtype=torch.uint8, device=dev)
        ds = torch.zeros(1 << 22, dtype=torch.uint8, device=dev)
        # odd map size: no phase split
        assert lib.wsi_pf_split_bytes(1, 7, 8, 64, planes) == 0
        assert lib.wsi_conv3x3_bn_act_split(buf.data_ptr(), out.data_ptr(), None, wp0.data_ptr(), b0.data_ptr(), 1, 7, 8, 64, 64, 1,
                                            planes, st()) == -22
        # output maps wider than 33 (input 80x80 -> 40x40): the wide stride-2 kernel declines, the caller keeps ordinary PF
        big = torch.zeros(lib.wsi_pf_split_bytes(1, 80, 80, 64, planes), dtype=torch.uint8, device=dev)
        assert lib.wsi_conv3x3s2_ds_fused_split(big.data_ptr(), out.data_ptr(), ds.data_ptr(), wp3.data_ptr(), b3.data_ptr(),
                                                wp1.data_ptr(), b1.data_ptr(), 1, 80, 80, 64, 128, planes, st()) == -22
        # aliasing output pointers
        assert lib.wsi_conv3x3s2_ds_fused_split(buf.data_ptr(), out.data_ptr(), out.data_ptr(), wp3.data_ptr(), b3.data_ptr(),
                                                wp1.data_ptr(), b1.data_ptr(), 1, 16, 16, 64, 128, planes, st()) == -22
    # single-plane (speed) mode has no phase-split path
    wp0, b0 = E.prepack_conv(w0, None, 1, dev)
    assert lib.wsi_conv3x3_bn_act_split(buf.data_ptr(), out.data_ptr(), None, wp0.data_ptr(), b0.data_ptr(), 1, 8, 8, 64, 64, 1, 1,
                                        st()) == -22
    torch.cuda.synchronize()


def test_random_shapes_sweep(dev, monkeypatch):
    """12 random (n, c, h, w) shapes - non-square maps, odd sizes - through the stride-1 kernels and the phase-split
    stride-2 block in both split modes (tests/studies/random_conv_shapes.py runs the same sweep at length)."""
    import os
    import runpy
    import sys
    path = os.path.join(os.path.dirname(__file__), 'studies', 'random_conv_shapes.py')
    monkeypatch.setattr(sys, 'argv', [path, '3', '12'])
    runpy.run_path(path, run_name='__main__')


@pytest.mark.parametrize('shape', [(37, 128, 32, 32), (5, 128, 32, 32), (70, 256, 16, 16), (3, 256, 16, 16), (130, 512, 8, 8),
                                   (9, 512, 8, 8), (2, 256, 2, 2), (3, 128, 64, 64), (1, 512, 1, 1)])
def test_pingpong_conv_equals_wide_kernel(dev, shape):
    """conv3x3s1_pp_kernel (cfg 70-72: 8-wave ping-pong schedule, double-buffered slab) is bit-identical to the slab3 kernel
    (cfg 30) in every precision mode, with and without residual, ragged last tiles and tiny maps included; pad
    positions stay untouched."""
    import ctypes as C
    from wsi_segmentation_pipeline_amd import native, engine as E
    lib = native.load()
    n, c, h, w = shape
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(n * 7 + c)
    x = torch.randn(n, c, h, w, generator=g).abs_()
    r = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
    ran = 0
    for planes in (3, 2, 1):
        wpk, bias = E.prepack_conv(wt, None, planes, dev)
        xpf, rpf = E.pf_pack(x.to(dev), planes), E.pf_pack(r.to(dev), planes)
        for resid in (None, rpf):
            for relu in (0, 1):
                ref = E.pf_zeros(n, c, h, w, planes, dev)
                native.check(lib.wsi_conv3x3_bn_act_cfg(xpf.data_ptr(), ref.data_ptr(), resid.data_ptr() if resid is not None else None,
                                                        wpk.data_ptr(), bias.data_ptr(), n, h, w, c, c, 1, relu, planes, 30, st), 'cfg 30')
                for cfg in (70, 71, 72, 73, 74, 77, 78):
                    out = E.pf_zeros(n, c, h, w, planes, dev)
                    rc = lib.wsi_conv3x3_bn_act_cfg(xpf.data_ptr(), out.data_ptr(), resid.data_ptr() if resid is not None else None,
                                                    wpk.data_ptr(), bias.data_ptr(), n, h, w, c, c, 1, relu, planes, cfg, st)
                    if rc == -22:
                        continue                                  # shape outside this tile configuration (LDS budget, cout multiple)
                    assert rc == 0
                    assert torch.equal(out, ref), (shape, planes, cfg, resid is not None, relu)
                    ran += 1
    assert ran > 0 or c < 128 or w > 33                        # (maps wider than 33: two slabs of this tile do not fit)


@pytest.mark.parametrize('shape', [(3, 64, 64), (1, 64, 4), (5, 64, 128), (2, 128, 64)])
def test_row_stacked_kernel_matches_slab3(dev, shape):
    """conv3x3s1_rows_kernel (cfg 40 / 41, r04: a wave's four pixel tiles are the same 32 columns of four map rows, so one set of
    pixel fragments serves up to three (tile, tap) steps) against conv3x3s1_slab3_kernel (cfg 38) on 64-wide maps, mode 3, with /
    without residual and ReLU (the 96-byte line route is covered by the trunk tests).  The two accumulate the same
    products in another order ((line, dx, dy) instead of (line, dy, dx)): fp32 sums agree to a few ulps of the largest partial sum,
    the stored fp16 + fp6 lines therefore to one fp6 step (2^-15 of the block maximum) on rare values; pad positions stay untouched."""
    import ctypes as C
    from wsi_segmentation_pipeline_amd import native, engine as E
    lib = native.load()
    n, c, h = shape
    w, co = 64, 64
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(n * 11 + h)
    x = torch.randn(n, c, h, w, generator=g).abs_()
    r = torch.randn(n, co, h, w, generator=g)
    wt = torch.randn(co, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
    wpk, bias = E.prepack_conv(wt, None, 3, dev)
    xpf, rpf = E.pf_pack(x.to(dev), 3), E.pf_pack(r.to(dev), 3)
    real = E.pf_pack(torch.ones(n, co, h, w).to(dev), 3).view(-1, 128)[:, :64].ne(0).any(1)
    for resid in (None, rpf):
        for relu in (0, 1):
            ref = E.pf_zeros(n, co, h, w, 3, dev)
            native.check(lib.wsi_conv3x3_bn_act_cfg(xpf.data_ptr(), ref.data_ptr(), resid.data_ptr() if resid is not None else None,
                                                    wpk.data_ptr(), bias.data_ptr(), n, h, w, c, co, 1, relu, 3, 38, st), 'cfg 38')
            rf = E.pf_unpack(ref, n, co, h, w, 3)
            for cfg in (40, 41):
                out = E.pf_zeros(n, co, h, w, 3, dev)
                native.check(lib.wsi_conv3x3_bn_act_cfg(xpf.data_ptr(), out.data_ptr(), resid.data_ptr() if resid is not None else None,
                                                        wpk.data_ptr(), bias.data_ptr(), n, h, w, c, co, 1, relu, 3, cfg, st), 'cfg %d' % cfg)
                got = E.pf_unpack(out, n, co, h, w, 3)
                err = float((got - rf).abs().max() / rf.abs().max())
                print('rows cfg %d vs slab3: resid=%s relu=%d max rel diff %.2e, %.4f %% of the values differ' %
                      (cfg, resid is not None, relu, err, 100.0 * float((got != rf).float().mean())))
                assert err <= 6.2e-5, (shape, cfg, err)               # one fp6 step of a block whose maximum is the tensor's (2^-14)
                assert not bool(out.view(-1, 128)[~real].ne(0).any()), 'kernel wrote to a pad position'


@pytest.mark.parametrize('shape', [(8, 8192, 4096), (70, 8192, 4096), (200, 512, 128), (64, 64, 256), (65, 96, 128), (2001, 4096, 4), (33, 128, 1),
                                   (100, 512, 16), (40, 512, 9)])
def test_linear_mfma_gemm(dev, shape):
    """The bag head fc.0 = Linear(8192 -> 4096) + ReLU on the fp32-input MFMA GEMM (exact fp32 products) and its second layer
    Linear(4096 -> 4) on the wave-per-row kernel, vs torch fp32 on the CPU; ragged bag counts, with and without bias / ReLU."""
    import ctypes as C
    from wsi_segmentation_pipeline_amd import native
    lib = native.load()
    b, k, j = shape
    g = torch.Generator().manual_seed(b + k)
    x, w, bias = torch.randn(b, k, generator=g), torch.randn(j, k, generator=g) * (1.0 / k ** 0.5), torch.randn(j, generator=g)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    xd, wd, bd = x.to(dev), w.to(dev), bias.to(dev)
    for relu, bb in ((1, bd), (0, None)):
        y = torch.full((b, j), float('nan'), device=dev)
        native.check(lib.wsi_linear(xd.data_ptr(), wd.data_ptr(), bb.data_ptr() if bb is not None else None, y.data_ptr(), b, k, j, relu, st), 'linear')
        ref = F.linear(x, w, bias if bb is not None else None)
        ref = F.relu(ref) if relu else ref
        assert float((y.cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_paint_regions_last_wins(dev):
    """wsi_paint_regions == the reference's sequential `pred_mask[foreground_indices] = cls` loop, overlapping regions included."""
    from wsi_segmentation_pipeline_amd import bags as B
    rng = np.random.default_rng(3)
    shape = (97, 131)
    lists, cls = [], rng.integers(0, 4, 40).astype(np.uint8)
    for r in range(40):
        if r % 3 == 0:
            m = np.zeros(shape, bool)
            y, x = rng.integers(0, 80), rng.integers(0, 110)
            m[y:y + rng.integers(1, 30), x:x + rng.integers(1, 30)] = True
            lists.append(np.nonzero(m))                              # tuple of index arrays, as the reference stores them
        else:
            lists.append(rng.integers(0, shape[0] * shape[1], rng.integers(0, 200)))
    ref = np.zeros(shape, np.int64)
    for idx, c in zip(lists, cls):
        if isinstance(idx, tuple):
            ref[idx] = c
        else:
            ref.reshape(-1)[idx] = c
    got = B.paint_regions(shape, lists, torch.from_numpy(cls).to(dev), dev)
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), ref)
    assert B.paint_regions(shape, [], torch.zeros(0, dtype=torch.uint8, device=dev), dev).sum().item() == 0


def test_stitch_exponent_guard(dev):
    """wsi_exponent_span + check_stitch_exact: the float64 atomics are exact (order-independent) only inside a bounded exponent
    span; the guard measures it on the device and the driver refuses beyond it."""
    from wsi_segmentation_pipeline_amd import engine as E
    v = torch.tensor([0.0, 1.5, -3.0, 1e-3, float('inf'), float('nan'), 100.0], device=dev)
    lo, hi = (int(t) for t in E.exponent_span(v).cpu())
    assert (lo, hi) == (127 - 10, 127 + 6)                           # 1e-3 = 1.024 x 2^-10, 100 = 1.5625 x 2^6
    E.check_stitch_exact(E.exponent_span(v), 16)                       # 16 + 4 bits: fine
    wide = torch.tensor([1e-12, 1e3], device=dev)
    with pytest.raises(RuntimeError):
        E.check_stitch_exact(E.exponent_span(wide), 4)
    assert [int(t) for t in E.exponent_span(torch.zeros(5, device=dev)).cpu()] == [255, 0]
    E.check_stitch_exact(E.exponent_span(torch.zeros(5, device=dev)), 1000)


def test_mx_clamp_to_fp16_range(dev):
    """The documented deviation of the two split-precision modes: activations are clamped to the fp16 range (+-65504) when a conv
    writes its output lines (mode 3 since r01; mode 2 since its operand pair became fp16 in r05).  Centre-tap permutation weights of
    gain 300 on inputs up to 400 put outputs at up to 120000."""
    from wsi_segmentation_pipeline_amd import engine as E
    n, c, h, w = 2, 64, 8, 8
    g = torch.Generator().manual_seed(5)
    x = torch.rand(n, c, h, w, generator=g) * 400.0
    wt = torch.zeros(c, c, 3, 3)
    for co in range(c):
        wt[co, (co * 7 + 3) % c, 1, 1] = 300.0 if co % 2 == 0 else -300.0
    ref = F.conv2d(x, wt, None, 1, 1)
    assert float(ref.abs().max()) > 100000
    out = {}
    for planes in (2, 3):
        wpk, bias = E.prepack_conv(wt, None, planes, dev)
        out[planes] = E.pf_unpack(E.conv_bn_act(E.pf_pack(x.to(dev), planes), n, h, w, c, c, wpk, bias, 1, 3, None, False, planes),
                                  n, c, h, w, planes).cpu()
    clamped = ref.clamp(-65504.0, 65504.0)
    # parity (fp16 pair since r05; the bf16 pair of r01-r04 did not clamp): the same documented fp16-range saturation as mx
    assert float(out[2].abs().max()) == 65504.0
    assert float((out[2] - clamped).abs().max() / 65504.0) <= TOL_PARITY
    assert float(out[3].abs().max()) == 65504.0
    assert float((out[3] - clamped).abs().max() / 65504.0) <= TOL_MX
    inside = ref.abs() < 60000
    assert float((out[3][inside] - ref[inside]).abs().max() / 65504.0) <= TOL_MX
