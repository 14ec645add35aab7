"""GPU parity of the dense 'seg' model (ResNet-18 encoder + smp-style U-Net decoder on HIP kernels) against the torch-fp32 CPU
spec oracle/unet_oracle.py.  The decoder belongs to a third-party package the reference only calls (absent, un-pinned):
parity is unpinned by the reference, the spec is self-authored from the published architecture (oracle/unet_oracle.py header);
the encoder half is pinned through resnet_oracle.  The default mode (planes 2) is held to north_star's ABSOLUTE 1e-3 on logits of
magnitude 16 (r05); mx / speed are reported against looser bounds and not claimed on the dense path."""
import os

import numpy as np
import pytest
import torch

from oracle import resnet_oracle as R
from oracle import unet_oracle as U
from oracle import weights as W
from oracle import wsi_oracle as WO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def sd():
    return W.make_unet_state_dict(7, 4)


def _scaled_to_logit(sd, x, target=16.0):
    """The seeded decoder ends in |logit| of a few hundred (softmax saturated everywhere).  The 1e-3 contract of north_star is ABSOLUTE on
    logits of the size trained heads produce, so the final 1x1 conv is scaled until max |logit| of the spec == `target` on these inputs
    (bench.py --workload seg does the same with 8 / 216) - the magnitude of the hot precision-margin families of the cls path."""
    with torch.no_grad():
        s = target / float(U.unet_forward(sd, x).abs().max())
    sd = dict(sd)
    for key in ('decoder.final_conv.weight', 'decoder.final_conv.bias'):
        sd[key] = sd[key] * s
    return sd


# planes 2 (the drop-in default of UNetSeg): ABSOLUTE 1e-3 on per-pixel logits at |logit| = 16 - the contract.  mx / speed are reported
# against looser bounds and are not claimed on the dense path (r04: mx 2.2e-3 at |logit| 16).
@pytest.mark.parametrize('planes,tol,enc_tol', [(2, 1e-3, 5e-5), (3, 4.8e-2, 3e-3), (1, 1.0, 6e-2)])
def test_unet_forward_matches_oracle(dev, sd, planes, tol, enc_tol):
    from wsi_segmentation_pipeline_amd.unet import UNetEngine
    u8 = W.make_u8_patches(41, (3, 3, 64, 96))
    x = R.normalize_u8(u8)
    sd = _scaled_to_logit(sd, x)
    with torch.no_grad():
        ref = U.unet_forward(sd, x)
        enc_sd = {k[8:]: v for k, v in sd.items() if k.startswith('encoder.')}
        ref_enc = U.encoder(enc_sd, x)
    eng = UNetEngine(sd, dev, planes=planes)
    got, enc = eng.forward_f32(x.to(dev), logits=True, enc=True)
    scale = float(ref.abs().max())
    assert abs(scale - 16.0) < 1e-3
    err = float((got.cpu() - ref).abs().max())
    enc_err = [float((a.cpu() - b).abs().max() / b.abs().max()) for a, b in zip(enc, ref_enc)]
    print('planes %d: max |logit - spec| %.2e ABSOLUTE at max |logit| %.1f; encoder maps rel err %s' % (planes, err, scale, ['%.1e' % e for e in enc_err]))
    assert got.shape == ref.shape == (3, 4, 64, 96)
    assert err <= tol and max(enc_err) <= enc_tol
    # decoder alone on the oracle's encoder maps == the generic `model.decoder(model.encoder(x))` calling sequence
    dec = eng.decode([t.to(dev) for t in ref_enc])
    assert float((dec.cpu() - ref).abs().max()) <= tol
    # tiles read from a u8 slide (fused read + transform in the stems) == the f32 path
    strip = np.ascontiguousarray(u8.transpose(0, 2, 3, 1).reshape(-1, 96, 3))
    xy = np.stack((np.zeros(3, np.int32), np.arange(3, dtype=np.int32) * 64), 1)
    t = eng.forward_tiles(torch.from_numpy(strip).to(dev), torch.from_numpy(xy), 64, 96)
    assert float((t.cpu() - ref).abs().max()) <= tol
    # argmax agreement of the class maps (what the drivers consume)
    assert float((got.cpu().argmax(1) != ref.argmax(1)).float().mean()) <= (0.002 if planes >= 2 else 0.05)


def test_dense_logits_hold_the_absolute_contract_at_bench_shape(dev):
    """The bench's own dense workload (bench.py --workload seg: decoder seed 5, final conv x 8/216, random u8 tiles of 256 x 256,
    max |logit| ~ 16) in the drop-in's default mode: max |logit - spec| <= 1e-3 ABSOLUTE per pixel (r04's bf16 pair measured 1.011e-3
    here and the relative tolerance of the r04 test did not see it; the fp16 pair of r05 is expected near 1e-4)."""
    from wsi_segmentation_pipeline_amd.unet import UNetEngine
    usd = W.make_unet_state_dict(5, classes=4)
    for key in ('decoder.final_conv.weight', 'decoder.final_conv.bias'):
        usd[key] = usd[key] * (8.0 / 216.0)
    g = torch.Generator(device=dev).manual_seed(1)
    level0 = torch.randint(0, 256, (512, 512, 3), dtype=torch.uint8, device=dev, generator=g)
    xy = torch.tensor([[0, 0], [256, 0], [0, 256], [256, 256]], dtype=torch.int32, device=dev)
    eng = UNetEngine(usd, dev, planes=2)
    got = eng.forward_tiles(level0, xy, 256, 256).cpu()
    l0 = level0.cpu().numpy()
    u8 = np.stack([l0[y:y + 256, x:x + 256] for x, y in xy.cpu().numpy()]).transpose(0, 3, 1, 2)
    with torch.no_grad():
        ref = U.unet_forward(usd, R.normalize_u8(u8))
    err, scale = float((got - ref).abs().max()), float(ref.abs().max())
    print('dense path, parity (fp16 pair): max |logit - spec| %.2e at max |logit| %.1f' % (err, scale))
    assert 8.0 < scale < 40.0
    assert err <= 1e-3
    assert err <= 7e-4, 'round-5 target: 0.7 of the contract'


def test_unet_256_batch_and_module_surface(dev, sd):
    """256x256 tiles (the bench shape), ragged batches, and the nn.Module surface the reference drives: state-dict round trip,
    model(x), model.encoder(x) -> five maps deepest first, model.decoder(encoding)."""
    from wsi_segmentation_pipeline_amd.unet import UNetSeg
    model = UNetSeg(4)
    model.load_state_dict(sd, strict=True)
    assert set(model.state_dict().keys()) == set(sd.keys())
    model = model.cuda().eval()
    u8 = W.make_he_patches(5, 3, 256)
    x = R.normalize_u8(u8)
    with torch.no_grad():
        ref = U.unet_forward(sd, x)
        y = model(x.cuda())
        enc = model.encoder(x.cuda())
        y2 = model.decoder(enc)
    scale = float(ref.abs().max())
    assert [tuple(t.shape[1:]) for t in enc] == [(512, 8, 8), (256, 16, 16), (128, 32, 32), (64, 64, 64), (64, 128, 128)]
    assert float((y.cpu() - ref).abs().max()) / scale <= 2e-4
    assert float((y2.cpu() - ref).abs().max()) / scale <= 4e-4          # (encoder maps went through fp32 and were re-packed)
    with pytest.raises(RuntimeError):
        model(x)                                                      # CPU tensor: no fallback


def test_predict_tumorbed_seg_and_predict_wsis_with_unet(dev, sd, tmp_path):
    """The reference's default mode: predict_tumorbed(mode='seg') (decoder blocks stitched at the map's own level) and
    predict_wsis driving the U-Net, both against the CPU oracle chain (U-Net spec -> float64 stitch -> threshold / heat map,
    tumour-bed post-process)."""
    import myargs
    import utils.dataset as ds
    import utils.eval as val
    from oracle import postprocess_oracle as P
    from wsi_segmentation_pipeline_amd.slide import ArraySlide
    from wsi_segmentation_pipeline_amd.unet import UNetSeg
    a = myargs.args
    a.scan_level, a.scan_resize, a.num_classes, a.class_probs = 2, 1, 4, [0., 0., 0., 0.]
    a.tile_w = a.tile_h = 64
    a.tile_stride_w = a.tile_stride_h = 48
    a.val_save_pth, a.wsi_mask_pth = str(tmp_path / 'out'), str(tmp_path / 'nomask')
    rng = np.random.default_rng(13)
    l2 = np.clip(np.kron(rng.integers(60, 250, (7, 9, 3)), np.ones((32, 32, 1))) + rng.integers(-25, 25, (224, 288, 3)), 0, 255).astype(np.uint8)
    big = np.repeat(np.repeat(l2, 4, 0), 4, 1)
    slide = ArraySlide([np.repeat(np.repeat(big, 4, 0), 4, 1)[:8, :8], big[:8, :8], l2], [1.0, 4.0, 16.0])   # only level 2 is read
    slide.level_dimensions = ((288 * 16, 224 * 16), (288 * 4, 224 * 4), (288, 224))
    slide.name = 'seg.svs'
    model = UNetSeg(4)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    model.classifier, model.regressor = torch.nn.Identity(), torch.nn.Identity()
    params = {'ph': 64, 'pw': 64, 'sh': 48, 'sw': 48}
    dataset = ds.Dataset_wsis({'seg.svs': slide}, params, bs=5)
    tiles = dataset.wsis['seg.svs']['iterator'].dataset.datalist
    mask = dataset.wsis['seg.svs']['mask']
    assert len(tiles) >= 12
    res = val.predict_tumorbed(model, dataset, 1, mode='seg')['seg.svs']
    u8 = np.stack([WO.read_tile(l2, x, y, 64, 64) for x, y in tiles]).transpose(0, 3, 1, 2)
    with torch.no_grad():
        tp = U.unet_forward(sd, R.normalize_u8(u8)).numpy()
    pred = WO.stitch_tumorbed(tiles, tp, 4, l2.shape[:2], 1.0, 64, 64)
    ref_cls, ref_probs = WO.threshold_probs(pred)
    ref_heat = WO.tumorbed_heatmap(ref_probs, mask, 'seg')
    flips = float((res['classes'] != ref_cls).mean())
    dheat = np.abs(res['heatmap'].astype(int) - ref_heat.astype(int))
    print('seg: class flips %.4f, heat pixels off by > 1: %d of %d' % (flips, int((dheat > 1).sum()), dheat.size))
    assert flips <= 0.002 and (dheat > 1).mean() <= 0.002            # logits differ by ~1e-4 relative: a few ties may flip
    assert os.path.exists('%s/1/seg.svs_48_heatmap.png' % a.val_save_pth)
    # predict_wsis with the same model: accumulate at scan level, argmax, tumour bed
    dataset = ds.Dataset_wsis({'seg.svs': slide}, params, bs=5)
    out = val.predict_wsis(model, dataset, 2)['seg.svs']
    ref_w = WO.stitch_wsis(tiles, tp, 4, l2.shape[:2], 64, 64)
    assert float(np.abs(out['pred'].cpu().numpy() - ref_w).max()) <= 3e-4 * float(np.abs(ref_w).max())
    p = out['classes_level2'].cpu().numpy()
    tb_pred, outline = P.tumor_bed(p)                                # post-process: bit-exact given the device's own class map
    assert np.array_equal(out['tumor_bed'].cpu().numpy(), tb_pred) and np.array_equal(out['outline'].cpu().numpy(), outline)
    assert float((p != np.argmax(ref_w, 0)).mean()) <= 0.002


@pytest.mark.parametrize('shape', [(3, 64, 64, 32, 16, 24), (2, 32, 0, 32, 40, 72), (2, 256, 128, 128, 8, 8), (2, 256, 128, 128, 32, 32), (1, 128, 64, 64, 12, 20),
                                   (5, 512, 256, 256, 4, 4)])
def test_fused_upsample_concat_conv_is_bit_identical(dev, shape):
    """r04: the first conv of a decoder block reads the low-resolution tensor and the skip directly (nearest x2 upsample + concat as
    source addresses of its slab DMA, wsi_conv3x3_up_concat_bn_act) - the same bits as materialising cat(up2(x), skip) first and
    running the same kernel on it (the copy moved whole 128-byte lines: no arithmetic either way); pad positions stay untouched.
    Shapes: (n, c_up, c_skip, cout, h, w) of the OUTPUT map - ragged tiles, no skip (the last block), 32 / 64 / 128 / 256 couts."""
    import ctypes as C
    import torch.nn.functional as F
    from wsi_segmentation_pipeline_amd import native, engine as E
    lib = native.load()
    n, cu, cs, co, h, w = shape
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(n * 13 + cu + h)
    xu = torch.randn(n, cu, h // 2, w // 2, generator=g).abs_()
    xs = torch.randn(n, cs, h, w, generator=g).abs_() if cs else None
    wt = torch.randn(co, cu + cs, 3, 3, generator=g) * (2.0 / (9 * (cu + cs))) ** 0.5
    cat = F.interpolate(xu, scale_factor=2, mode='nearest')
    if cs:
        cat = torch.cat([cat, xs], 1)
    ran = 0
    for planes in (3, 2):
        wpk, bias = E.prepack_conv(wt, None, planes, dev)
        up_pf = E.pf_pack(xu.to(dev), planes)
        sk_pf = E.pf_pack(xs.to(dev), planes) if cs else None
        cat_pf = E.pf_pack(cat.to(dev), planes)
        out = E.pf_zeros(n, co, h, w, planes, dev)
        rc = lib.wsi_conv3x3_up_concat_bn_act(up_pf.data_ptr(), sk_pf.data_ptr() if cs else None, out.data_ptr(), wpk.data_ptr(), bias.data_ptr(),
                                              n, h, w, cu, cs, co, 1, planes, st)
        assert rc == 0, (shape, planes, rc)
        # the unfused route on a slab3 configuration (every stride-1 kernel but the row-stacked one gives the same bits)
        ref = E.pf_zeros(n, co, h, w, planes, dev)
        cfg = 30 if co % 128 == 0 else (31 if co % 64 == 0 else 90)
        rc = lib.wsi_conv3x3_bn_act_cfg(cat_pf.data_ptr(), ref.data_ptr(), None, wpk.data_ptr(), bias.data_ptr(), n, h, w, cu + cs, co, 1, 1, planes, cfg, st)
        if rc == -22 and cfg == 90:
            rc = lib.wsi_conv3x3_bn_act_cfg(cat_pf.data_ptr(), ref.data_ptr(), None, wpk.data_ptr(), bias.data_ptr(), n, h, w, cu + cs, co, 1, 1, planes, 91, st)
        assert rc == 0, (shape, planes, cfg, rc)
        assert torch.equal(out, ref), (shape, planes)
        got = E.pf_unpack(out, n, co, h, w, planes)
        want = F.relu(F.conv2d(cat, wt, None, 1, 1))
        assert float((got.cpu() - want).abs().max() / want.abs().max()) <= (2e-3 if planes == 3 else 2e-4)
        ran += 1
    assert ran == 2


@pytest.mark.parametrize('shape', [(3, 64, 64), (2, 128, 192), (1, 256, 256), (5, 96, 128), (2, 64, 96)])
def test_fused_decoder_tail_agrees_with_the_three_launch_path_and_the_spec(dev, sd, shape):
    """r05 (csrc/tail.hip): parity mode runs the LAST decoder block and the 1x1 head as one kernel - the 3x3 conv on the upsampled map as
    a polyphase filter with host-summed weights, both intermediate tensors in LDS.  Against the three-launch path (A/B switch
    wsi_conv_set_mode +2097152) the logits differ by the fp32 rounding of the summed weights only; against the fp32 spec both hold the
    absolute 1e-3 contract at |logit| 16.  Shapes: one / several bands per image (a 256-row image of a small batch splits into 16 bands:
    every band seam recomputes its neighbours' conv1 rows), maps 64 to 256 wide (2 to 8 waves per workgroup), and a width whose half is
    not a multiple of 32 (96: the tail is refused, the three launches run - bit-identical)."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.unet import UNetEngine
    n, h, w = shape
    lib = native.load()
    u8 = W.make_u8_patches(90 + h + w, (n, 3, h, w))
    x = R.normalize_u8(u8)
    sd = _scaled_to_logit(sd, x[:1] if h * w > 128 * 192 else x)
    eng = UNetEngine(sd, dev, planes=2)
    assert eng.dw.tail_w
    fused = eng.forward_f32(x.to(dev))[0]
    try:
        lib.wsi_conv_set_mode(1 + 4194304)                                # the first form of the kernel (all waves do both convs)
        form1 = eng.forward_f32(x.to(dev))[0]
        lib.wsi_conv_set_mode(1 + 2097152)
        plain = eng.forward_f32(x.to(dev))[0]
    finally:
        lib.wsi_conv_set_mode(1)
    assert torch.equal(fused, eng.forward_f32(x.to(dev))[0])             # (the specialised form synchronises through an LDS counter: same bits every run)
    d12 = float((fused - form1).abs().max())
    assert d12 <= 2e-6 * max(float(plain.abs().max()), 1.0), d12          # same weights, same products; the head sums in another order
    assert torch.isfinite(fused).all()
    d = float((fused - plain).abs().max())
    scale = float(plain.abs().max())
    print('fused tail vs three launches %s: max |dlogit| %.3g at max |logit| %.3g' % (shape, d, scale))
    if (w // 2) % 32:
        assert torch.equal(fused, plain)
        return
    assert 0 < d <= 2e-5 * max(scale, 1.0)                                # another rounding of the same conv, not the same bits
    with torch.no_grad():
        ref = U.unet_forward(sd, x[:1])
    assert float((fused[:1].cpu() - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()) / 16.0)


@pytest.mark.parametrize('shape', [(3, 64, 64), (2, 128, 192), (3, 256, 256), (2, 96, 160)])
def test_stem_kernel_stores_the_half_resolution_skip_itself(dev, sd, shape):
    """r05: on the product path (tiles read from a u8 slide, parity mode) the fused stem + pool kernel also stores x0 = relu(bn1(conv1(x))),
    the conv values it pools anyway (exact integer products), where r02-r04 ran a second, unfused fp16-pair stem conv for it
    (A/B: wsi_conv_set_mode +8388608).  The logits of both routes agree to the pair's rounding and hold the contract against the spec;
    shapes: one strip (64-wide: even lane layout), several strips and segments, odd strip counts, tiles reaching past the slide."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.unet import UNetEngine
    n, h, w = shape
    lib = native.load()
    rng = np.random.default_rng(h + w)
    slide_np = rng.integers(0, 256, (h + 40, 2 * w + 24, 3), dtype=np.uint8)
    slide = torch.from_numpy(slide_np).to(dev)
    xy = torch.tensor([[0, 0], [w + 30, 44], [7, 13]][:n], dtype=torch.int32)       # the second tile reaches 6 / 4 pixels past the slide
    u8 = np.stack([WO.read_tile(slide_np, int(x), int(y), w, h) for x, y in xy]).transpose(0, 3, 1, 2)
    x = R.normalize_u8(u8)
    sd = _scaled_to_logit(sd, x[:1])
    eng = UNetEngine(sd, dev, planes=2)
    fused = eng.forward_tiles(slide, xy, h, w)
    lib.wsi_conv_set_mode(1 + 8388608)
    try:
        plain = eng.forward_tiles(slide, xy, h, w)
    finally:
        lib.wsi_conv_set_mode(1)
    d = float((fused - plain).abs().max())
    print('x0 from the pool kernel vs the unfused stem conv %s: max |dlogit| %.3g' % (shape, d))
    assert 0 < d <= 2e-4
    with torch.no_grad():
        ref = U.unet_forward(sd, x)
    assert float((fused.cpu() - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()) / 16.0)


@pytest.mark.parametrize('classes', [1, 2, 3])
def test_fused_decoder_tail_with_fewer_classes(dev, classes):
    """The fused tail pads the head to four classes in its weight blob and stores only the real ones: 1-3 classes against the three launches."""
    from wsi_segmentation_pipeline_amd import native
    from wsi_segmentation_pipeline_amd.unet import UNetEngine
    lib = native.load()
    sdc = W.make_unet_state_dict(9, classes)
    x = R.normalize_u8(W.make_u8_patches(77 + classes, (2, 3, 64, 128)))
    sdc = _scaled_to_logit(sdc, x)
    eng = UNetEngine(sdc, dev, planes=2)
    assert eng.dw.tail_w
    fused = eng.forward_f32(x.to(dev))[0]
    assert tuple(fused.shape) == (2, classes, 64, 128)
    lib.wsi_conv_set_mode(1 + 2097152)
    try:
        plain = eng.forward_f32(x.to(dev))[0]
    finally:
        lib.wsi_conv_set_mode(1)
    d = float((fused - plain).abs().max())
    assert 0 < d <= 2e-5 * max(float(plain.abs().max()), 1.0), d
    with torch.no_grad():
        ref = U.unet_forward(sdc, x)
    assert float((fused.cpu() - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()) / 16.0)
