"""CPU tests of oracle/postprocess_oracle.py (the spec the device post-process is held to).  The third-party calls the
reference makes (OpenCV / scikit-image / mahotas) are absent here, so the spec is checked against INDEPENDENT
implementations of the same published algorithms: SciPy's Qhull convex hull + half-plane test, brute-force
morphology, scipy.ndimage for the perimeter, and hand-derived cases."""
import numpy as np
import pytest

from oracle import postprocess_oracle as P


def _hull_scipy(img):
    from scipy.spatial import ConvexHull
    rr, cc = np.nonzero(img)
    pts = np.stack([rr, cc], 1).astype(float)
    off = np.array([[-.5, 0], [.5, 0], [0, -.5], [0, .5]])               # skimage _offsets_diamond(2)
    pts = (pts[:, None, :] + off).reshape(-1, 2)
    h = ConvexHull(pts)
    R, C = np.mgrid[:img.shape[0], :img.shape[1]]
    g = np.stack([R.ravel(), C.ravel()], 1).astype(float)
    return np.all(g @ h.equations[:, :2].T + h.equations[:, 2] <= 1e-9, axis=1).reshape(img.shape).astype(np.uint8)


def test_convex_hull_image_matches_qhull():
    rng = np.random.default_rng(0)
    for t in range(120):
        H, W = rng.integers(3, 40, 2)
        img = (rng.random((H, W)) < rng.choice([0.02, 0.1, 0.5])).astype(np.uint8)
        if t % 5 == 0:                                                    # one or two isolated pixels
            img[:] = 0
            img[rng.integers(0, H), rng.integers(0, W)] = 1
            if t % 10 == 0:
                img[rng.integers(0, H), rng.integers(0, W)] = 1
        if not img.any():
            assert not P.convex_hull_image(img).any()
            continue
        assert np.array_equal(P.convex_hull_image(img), _hull_scipy(img)), t
    assert P.convex_hull_image(np.zeros((5, 7), np.uint8)).sum() == 0
    # hand case: two pixels on a diagonal -> the hull of their diamonds covers exactly those two pixel centres
    img = np.zeros((4, 4), np.uint8)
    img[0, 0] = img[2, 2] = 1
    assert P.convex_hull_image(img).tolist() == [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 0]]


@pytest.mark.parametrize('k', [1, 3, 4, 5, 20])
def test_rect_morphology_matches_brute_force(k):
    rng = np.random.default_rng(k)
    img = (rng.random((53, 61)) < 0.8).astype(np.uint8)
    h = k // 2
    pad1 = np.pad(img, ((h, k - h - 1), (h, k - h - 1)), constant_values=1)
    pad0 = np.pad(img, ((h, k - h - 1), (h, k - h - 1)), constant_values=0)
    er, di = np.ones_like(img), np.zeros_like(img)
    for i in range(k):
        for j in range(k):
            er = np.minimum(er, pad1[i:i + 53, j:j + 61])                 # src(y + i - k//2, x + j - k//2), border ignored
            di = np.maximum(di, pad0[i:i + 53, j:j + 61])
    assert np.array_equal(P.erode_rect(img, k), er)
    assert np.array_equal(P.dilate_rect(img, k), di)
    assert np.array_equal(P.morph_open(img, k), P.dilate_rect(P.erode_rect(img, k), k))


def test_bwperim_matches_ndimage():
    from scipy import ndimage as ndi
    rng = np.random.default_rng(3)
    img = ndi.binary_dilation(rng.random((40, 50)) < 0.03, iterations=3)
    cross = ndi.generate_binary_structure(2, 1)
    ref = img & ~ndi.binary_erosion(img, cross, border_value=0)          # set, and a 4-neighbour (outside = 0) is not
    assert np.array_equal(P.bwperim(img), ref.astype(np.uint8))


def test_resize_bilinear_hand_cases():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 32, 48))
    assert np.array_equal(P.resize_bilinear(x, (32, 48)), x)             # same size: identity, bit for bit
    y = P.resize_bilinear(x, (2, 3))                                     # x16: the mean of the four centre pixels of each block
    for i in range(2):
        for j in range(3):
            blk = x[:, 16 * i + 7:16 * i + 9, 16 * j + 7:16 * j + 9]
            assert np.allclose(y[:, i, j], blk.mean((1, 2)), rtol=0, atol=1e-15)
    up = P.resize_bilinear(x[:, :4, :4], (8, 8))                         # upscale: edge-clamped
    assert np.array_equal(up[:, 0, 0], x[:, 0, 0]) and np.array_equal(up[:, -1, -1], x[:, 3, 3])


def test_tumor_bed_pipeline_and_scores():
    rng = np.random.default_rng(5)
    cls = np.zeros((120, 150), np.uint8)
    cls[30:80, 40:100] = 3
    cls[85:110, 20:45] = 2
    cls[rng.random(cls.shape) < 0.02] = 2                                # specks the 20x20 opening removes
    tb_pred, outline = P.tumor_bed(cls)
    opened = P.morph_open((cls >= 2).astype(np.uint8), 20)
    assert opened[50, 60] == 1 and opened[:25].sum() == 0
    assert tb_pred[82, 60] == 1                                          # between the two blobs: inside the hull
    assert (tb_pred >= opened).all() and outline.sum() > 0 and outline[50, 70] == 0
    gt = np.zeros_like(cls)
    gt[25:85, 35:105] = 3
    sc = P.wsi_scores(cls, gt, np.ones_like(cls))
    assert 0 < sc['acc'] <= 1 and sc['iou_fg'] > 0.5
    assert P.tumor_bed_iou(gt, tb_pred) == pytest.approx((gt.astype(bool) & tb_pred.astype(bool)).sum() /
                                                         (gt.astype(bool) | tb_pred.astype(bool)).sum(), abs=1e-9)
    poly = P.hull_polygon(opened)
    assert np.array_equal(poly[0], poly[-1]) and len(poly) >= 5
    pts = P.evenly_spaced_points_on_a_contour(poly, 32)
    assert pts.shape == (32, 2) and np.array_equal(pts[0], poly[0]) and np.allclose(pts[-1], poly[-1])


def test_scores_follow_the_reference_uint8_arithmetic():
    """utils/eval.py:110-111 evaluated literally on the dtypes the reference has - p = np.argmax (int64), gt = np.array(PIL image)
    (uint8) - so `1 - gt > 0` wraps to gt != 1.  Pixels with p == 0 and gt in {2, 3} carry weight 0 in the denominator of `s`
    (r02's int64 reading gave them weight 1)."""
    rng = np.random.default_rng(11)
    p = rng.integers(0, 4, (40, 50)).astype(np.int64)
    gt = rng.integers(0, 4, (40, 50)).astype(np.uint8)
    p[:10] = 0                                                           # plenty of p == 0 against gt = 2, 3
    gt[:10, :25] = 2
    gt[:10, 25:] = 3
    lit = 1 - np.sum(np.abs(p - gt)) / np.sum(np.maximum(np.abs(gt - 0), np.abs(gt - 3.0)) * (1 - (1 - (p > 0)) * (1 - gt > 0)))
    sc = P.wsi_scores(p, gt, np.ones_like(p))
    assert sc['s'] == pytest.approx(float(lit), abs=0, rel=1e-15)
    wrong = 1 - np.sum(np.abs(p - gt)) / np.sum(np.maximum(gt, np.abs(gt.astype(np.int64) - 3)) *
                                                (1 - (1 - (p > 0)) * (1 - gt.astype(np.int64) > 0)))
    assert abs(wrong - lit) > 1e-3                                       # the two readings really differ on this input
