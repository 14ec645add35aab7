"""GPU parity of the tumour-bed post-process (csrc/postproc.hip through the C ABI) against oracle/postprocess_oracle.py:
byte / index outputs bit-exact (counts of differing pixels asserted to be 0), float64 outputs bit-equal."""
import os

import numpy as np
import pytest
import torch

from oracle import postprocess_oracle as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _blobs(rng, h, w, n, rmax):
    yy, xx = np.mgrid[:h, :w]
    img = np.zeros((h, w), bool)
    for _ in range(n):
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(2, rmax)
        img |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    return img


@pytest.mark.parametrize('shape', [(97, 131), (256, 256), (64, 700), (5, 7)])
def test_morphology_hull_perimeter_match_oracle(dev, shape):
    from wsi_segmentation_pipeline_amd import postprocess as PP
    rng = np.random.default_rng(shape[0])
    h, w = shape
    img = (_blobs(rng, h, w, 6, max(3, min(h, w) // 3)) | (rng.random((h, w)) < 0.01)).astype(np.uint8)
    d = torch.from_numpy(img).to(dev)
    for k in (1, 3, 4, 20, 30):
        for op, ref in (('erode', P.erode_rect), ('dilate', P.dilate_rect), ('open', P.morph_open)):
            got = PP.morph_rect(d, k, op).cpu().numpy()
            assert np.array_equal(got, ref(img, k)), (k, op)
    assert np.array_equal(PP.bwperim(d).cpu().numpy(), P.bwperim(img))
    for src in (img, P.morph_open(img, 4), np.zeros_like(img), np.eye(h, w, dtype=np.uint8)):
        tb = PP.tumor_bed(torch.from_numpy(np.ascontiguousarray(src) * 3).to(dev), 2, 1, 1)        # open 1x1 / dilate 1x1 = identity
        assert np.array_equal(tb.tb_pred.cpu().numpy(), P.convex_hull_image(src))
        assert np.array_equal(tb.outline.cpu().numpy(), P.bwperim(P.convex_hull_image(src)))
        assert np.array_equal(tb.polygon().cpu().numpy(), P.hull_polygon(src))


def test_tumor_bed_full_size_cfg5(dev):
    """cfg5's post-process leg at the full 2500 x 2500 level-2 map: class map -> (>= 2) -> open 20 -> hull -> perimeter ->
    dilate 20 on the device == the oracle, zero differing pixels; the heat-map variant of paper_tools/overlay_tb_wsi.py too."""
    from wsi_segmentation_pipeline_amd import postprocess as PP
    rng = np.random.default_rng(17)
    h = w = 2500
    low = rng.random((h // 16 + 1, w // 16 + 1))
    field = np.kron(low, np.ones((16, 16)))[:h, :w]                       # tile-constant field, like a 'cls' map
    yy, xx = np.mgrid[:h, :w]
    bed = ((yy - 1300) / 700.0) ** 2 + ((xx - 1100) / 500.0) ** 2 < 1
    cls = np.where(bed & (field > 0.25), 3, np.where(field > 0.97, 2, (field > 0.5).astype(np.uint8))).astype(np.uint8)
    tb = PP.tumor_bed(torch.from_numpy(cls).to(dev), 2, 20, 20)
    ref_pred, ref_outline = P.tumor_bed(cls)
    n1 = int((tb.tb_pred.cpu().numpy() != ref_pred).sum())
    n2 = int((tb.outline.cpu().numpy() != ref_outline).sum())
    n3 = int((tb.opened.cpu().numpy() != P.morph_open((cls >= 2).astype(np.uint8), 20)).sum())
    print('cfg5 post-process at 2500x2500: %d hull / %d outline / %d opened pixels differ; hull area %d' % (n1, n2, n3, ref_pred.sum()))
    assert n1 == 0 and n2 == 0 and n3 == 0 and ref_pred.sum() > 1e5
    # scores and IoU: exact integer sums -> identical floats
    gt = np.where(bed, 3, 0).astype(np.uint8)
    mask = (rng.random((h, w)) < 0.9).astype(np.uint8)
    got = PP.wsi_scores(torch.from_numpy(cls).to(dev), torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev))
    ref = P.wsi_scores(cls, gt, mask)
    assert got == ref, (got, ref)
    assert PP.mask_iou(torch.from_numpy(bed.astype(np.uint8)).to(dev), tb.tb_pred) == P.tumor_bed_iou(bed, ref_pred)
    # heat-map variant
    heat = np.uint8(255 * np.clip(field * bed + 0.2 * field, 0, 1))
    tbh = PP.tumor_bed_from_heatmap(torch.from_numpy(heat).to(dev))
    im, hull, outl = P.tumor_bed_from_heatmap(heat)
    assert np.array_equal(tbh.opened.cpu().numpy(), im) and np.array_equal(tbh.tb_pred.cpu().numpy(), hull)
    assert np.array_equal(tbh.outline.cpu().numpy(), outl)
    # evenly spaced outline points: esp over the hull polygon, bit-equal with NumPy
    pts = tb.outline_points(64).cpu().numpy()
    assert np.array_equal(pts, P.evenly_spaced_points_on_a_contour(P.hull_polygon(P.morph_open((cls >= 2).astype(np.uint8), 20)), 64))


def test_esp_matches_reference_golden(dev, golden_dir):
    """contour_ordering.evenly_spaced_points_on_a_contour on the device vs the golden vectors generated from the
    reference itself (tests/golden/esp.npz) and bit-equal with the NumPy restatement."""
    from oracle import wsi_oracle as WO
    from wsi_segmentation_pipeline_amd import postprocess as PP
    g = np.load(os.path.join(golden_dir, 'esp.npz'))
    for src, n, key in (('contour', 16, 'esp16'), ('contour', 8, 'esp8'), ('square', 9, 'esp_sq9')):
        got = PP.esp(torch.from_numpy(g[src]).to(dev), n).cpu().numpy()
        assert np.abs(got - g[key]).max() <= 1e-12
        assert np.array_equal(got, WO.evenly_spaced_points_on_a_contour(g[src], n))
    rng = np.random.default_rng(2)
    for n in (2, 3, 50, 1000):
        c = np.cumsum(rng.standard_normal((n, 2)), 0)
        c[n // 2] = c[n // 2 - 1]                                        # a zero-length segment (duplicate abscissa)
        for num in (1, 2, 7, 333):
            got = PP.esp(torch.from_numpy(c).to(dev), num).cpu().numpy()
            assert np.array_equal(got, WO.evenly_spaced_points_on_a_contour(c, num)), (n, num)


def test_resize_and_argmax_match_oracle(dev):
    from wsi_segmentation_pipeline_amd import postprocess as PP
    rng = np.random.default_rng(4)
    x = rng.standard_normal((4, 160, 208))
    d = torch.from_numpy(x).to(dev)
    for hw in ((160, 208), (10, 13), (40, 52), (77, 33), (320, 400), (1, 1)):
        assert np.array_equal(PP.resize_bilinear(d, hw).cpu().numpy(), P.resize_bilinear(x, hw)), hw
    x[:, 3, 5] = 0.25                                                     # ties -> first maximum
    x[2, 7, 7] = np.nan
    got = PP.argmax_classes(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.array_equal(got, np.argmax(x, 0).astype(np.uint8))


def test_predict_wsis_postprocess_matches_oracle(dev, tmp_path):
    """predict_wsis end to end with a dense GPU module: accumulate -> resize to level 2 -> argmax -> tumour bed -> scores,
    every stage equal to the CPU oracle chain fed with the same per-tile predictions."""
    import myargs
    import utils.dataset as ds
    import utils.eval as val
    from oracle import resnet_oracle as R
    from oracle import wsi_oracle as WO
    from wsi_segmentation_pipeline_amd.slide import ArraySlide
    a = myargs.args
    a.scan_level, a.scan_resize, a.num_classes, a.class_probs = 0, 1, 4, [0., 0., 0., 0.]
    a.tile_w = a.tile_h = 64
    a.tile_stride_w = a.tile_stride_h = 48
    a.val_save_pth, a.wsi_mask_pth = str(tmp_path / 'out'), str(tmp_path / 'nomask')
    rng = np.random.default_rng(11)
    H, W = 1600, 2080
    yy, xx = np.mgrid[:H, :W]
    blob = ((yy - 800) / 500.0) ** 2 + ((xx - 1000) / 700.0) ** 2 < 1
    l0 = np.where(blob[..., None], np.array([150, 60, 160]), np.array([235, 200, 230])).astype(np.uint8)
    l0 = np.clip(l0 + rng.integers(-20, 20, l0.shape), 0, 255).astype(np.uint8)
    slide = ArraySlide([l0, l0[::4, ::4], l0[::16, ::16]], [1.0, 4.0, 16.0])
    slide.name = 'pp.svs'

    class Dense(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([[0.5, 0.25, -0.25], [-1.0, 0.5, 0.5], [1.0, -2.0, 1.0], [-0.5, -1.0, 2.0]]))

        def forward(self, x):                                     # (B,3,h,w) -> (B,4,h,w): exact in fp32 (dyadic weights, short sums)
            return torch.stack([sum(x[:, c] * float(self.w[k, c]) for c in range(3)) for k in range(4)], 1)

    model = Dense().cuda()
    dataset = ds.Dataset_wsis({'pp.svs': slide}, {'ph': 64, 'pw': 64, 'sh': 48, 'sw': 48}, bs=64)
    entry = dataset.wsis['pp.svs']
    map_hw = l0[::16, ::16].shape[:2]
    gt = np.where(blob[::16, ::16], 3, 0).astype(np.uint8)
    entry['gt'], entry['tb_gt'] = gt, (gt > 0).astype(np.uint8) * 255
    tiles = entry['iterator'].dataset.datalist
    mask = entry['mask']
    res = val.predict_wsis(model, dataset, 3)['pp.svs']
    # oracle chain
    u8 = np.stack([WO.read_tile(l0, x, y, 64, 64) for x, y in tiles]).transpose(0, 3, 1, 2)
    with torch.no_grad():
        tile_pred = Dense()(R.normalize_u8(u8)).numpy()
    ref = WO.stitch_wsis(tiles, tile_pred, 4, l0.shape[:2], 64, 64)
    assert np.array_equal(res['pred'].cpu().numpy(), ref)
    ref2 = P.resize_bilinear(ref, map_hw)
    assert np.array_equal(res['pred_level2'].cpu().numpy(), ref2)
    p = np.argmax(ref2, 0).astype(np.uint8)
    assert np.array_equal(res['classes_level2'].cpu().numpy(), p)
    tb_pred, outline = P.tumor_bed(p)
    assert np.array_equal(res['tumor_bed'].cpu().numpy(), tb_pred) and np.array_equal(res['outline'].cpu().numpy(), outline)
    sc = P.wsi_scores(p, gt, (np.asarray(mask) > 0).astype(np.uint8))
    sc['iou_tb'] = P.tumor_bed_iou(gt > 0, tb_pred)
    assert res['scores'] == sc, (res['scores'], sc)
    assert os.path.exists('%s/3/pp.svs_48.png' % a.val_save_pth)


def test_contour_ordering_dropin(dev, golden_dir):
    """The root-level `contour_ordering` module (the name the reference scripts import) returns the reference's golden
    outputs for array-likes and CUDA tensors."""
    import contour_ordering as co
    g = np.load(os.path.join(golden_dir, 'esp.npz'))
    got = co.evenly_spaced_points_on_a_contour(g['contour'], 16)
    assert isinstance(got, np.ndarray) and np.abs(got - g['esp16']).max() <= 1e-12
    got = co.evenly_spaced_points_on_a_contour(g['square'].tolist(), 9)
    assert np.abs(got - g['esp_sq9']).max() <= 1e-12
    t = co.evenly_spaced_points_on_a_contour(torch.from_numpy(g['contour']).to(dev), 8)
    assert t.is_cuda and np.abs(t.cpu().numpy() - g['esp8']).max() <= 1e-12


def test_scores_uint8_wrap_weight_term(dev):
    """wsi_score_counts against the literal uint8 NumPy evaluation of utils/eval.py:110-111 on a map with every (p, gt) pair in
    {0..3}^2, in particular p == 0 against gt = 2, 3 (weight 0 in the reference: `1 - gt` wraps)."""
    from wsi_segmentation_pipeline_amd import postprocess as PP
    rng = np.random.default_rng(23)
    p = rng.integers(0, 4, (64, 96)).astype(np.uint8)
    gt = rng.integers(0, 4, (64, 96)).astype(np.uint8)
    p[:16] = 0
    gt[:16, :48], gt[:16, 48:] = 2, 3
    mask = (rng.random(p.shape) < 0.8).astype(np.uint8)
    got = PP.wsi_scores(torch.from_numpy(p).to(dev), torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev))
    pi = p.astype(np.int64)                                                      # np.argmax result in the reference
    lit = lambda q: float(1 - np.sum(np.abs(q - gt)) / np.sum(np.maximum(np.abs(gt - 0), np.abs(gt - 3.0)) * (1 - (1 - (q > 0)) * (1 - gt > 0))))
    assert got['s'] == lit(pi) and got['s_masked'] == lit(mask.astype(np.int64) * pi)
    assert got == P.wsi_scores(p, gt, mask)
